// radix_sort_basic_kdf.hpp -- key-derivation functions of the MI355X radix sort's C++ surface.
//
// Same public names and results as the reference's radix_sort_basic_kdf.hpp (namespace basic_kdfs,
// kdf(value), highbit<T>()), so code written against the reference compiles unchanged:
//   unsigned integers  -> the value itself                       (reference :19-23)
//   signed integers    -> value ^ highbit<T>(), as unsigned      (reference :26-30)
//   float / double     -> bits ^ (-(bits >> 31|63) | highbit)    (reference :32-46)
// bool is not a key type (reference :20,:27).  These host functions are what an opaque user KeyFunc
// composes with; for plain scalar arrays the same arithmetic runs on the GPU (rsx_kernels.hpp, kdf_apply).
#pragma once

#include <cstdint>
#include <cstring>
#include <type_traits>

namespace basic_kdfs {

template <typename T>
constexpr std::enable_if_t<std::is_integral_v<T>, std::make_unsigned_t<T>> highbit()
{
	return static_cast<std::make_unsigned_t<T>>(std::make_unsigned_t<T>(1) << (sizeof(T) * 8 - 1));
}

namespace detail {
template <typename T> struct bits_of;
template <> struct bits_of<float> { using type = uint32_t; };
template <> struct bits_of<double> { using type = uint64_t; };
template <typename T> inline constexpr bool is_key_scalar_v =
	(std::is_integral_v<T> && !std::is_same_v<T, bool>) || std::is_same_v<T, float> || std::is_same_v<T, double>;
}  // namespace detail

// One template instead of the reference's overload set; kdf<T> is still a plain function, so
// decltype(basic_kdfs::kdf<T>) names its type exactly as in the reference's default template argument.
template <typename T>
auto kdf(const T &value)
{
	static_assert(detail::is_key_scalar_v<T>, "basic_kdfs::kdf: integer (not bool), float or double keys only");
	if constexpr (std::is_integral_v<T> && std::is_unsigned_v<T>) {
		return value;
	} else if constexpr (std::is_integral_v<T>) {
		using U = std::make_unsigned_t<T>;
		return static_cast<U>(static_cast<U>(value) ^ highbit<T>());
	} else {
		using U = typename detail::bits_of<T>::type;
		U u;
		std::memcpy(&u, &value, sizeof(u));
		const U top = U(1) << (sizeof(U) * 8 - 1);
		return static_cast<U>(u ^ ((u & top) ? ~U(0) : top));
	}
}

}  // namespace basic_kdfs
