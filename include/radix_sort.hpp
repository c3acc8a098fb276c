// radix_sort.hpp -- drop-in for the reference's header of the same name, backed by librsx.so (MI355X).
//
// Same template surface (reference radix_sort.hpp:98-99 and :31-32):
//
//     T* radix_sort(T* src, T* aux, size_t n, KeyFunc&& kf = basic_kdfs::kdf)
//     T* rs_sort_main(T* src, T* aux, size_t n, Hist& histogram, KeyFunc&& kf = basic_kdfs::kdf)
//
// and the same observable contract: stable order by kf(element), n < 2 and
// pre-sorted inputs return src with aux untouched, otherwise the result is in src when the number of
// non-constant 8-bit key columns is even and in aux when it is odd; callers use the returned pointer.
//
// Where the work happens
//   * T a scalar and kf the default basic_kdfs::kdf (or rsx_kdf::descending<T>): rsx_sort() -- the
//     keys are sorted on the GPU with the key derivation done in the kernels.
//   * rsx_kdf::by_member<&T::field[, descending]> on records: rsx_sort_records_tagged() -- key extraction, rank
//     sort and the gather of the records on the GPU.
//   * any other callable (radix_tests.cpp:41-43,:111-113,:175-177 shapes): kf is evaluated once per
//     element on the host into an array of KeyType, the GPU rank-sorts those keys and gathers the
//     (trivially copyable) elements: rsx_sort_records().  Elements that are movable but NOT trivially
//     copyable (the reference only ever move-assigns T, radix_sort.hpp:85-87): the GPU rank-sorts the
//     keys (rsx_sort_rank_keys) and the host moves the elements into that order, through `aux` as the
//     reference does, so that the result lies where its parity rule says.
// There is no CPU sorting path: without a usable MI355X the call throws std::runtime_error, where the
// reference "cannot fail".  The buffer that is not returned has unspecified contents (the reference
// leaves the previous pass's data there).
#pragma once

#include <cstddef>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <utility>
#include <vector>

#include "radix_sort_basic_kdf.hpp"
#include "rsx.h"

#ifndef RESTRICT
#define RESTRICT __restrict__
#endif

namespace rsx_kdf {
// Tagged descending KDF: the complement of basic_kdfs::kdf (README.md:564-574), recognised by the
// wrapper so that descending scalar sorts also stay on the device.
template <typename T> struct descending {
	auto operator()(const T &v) const { return static_cast<decltype(basic_kdfs::kdf<T>(v))>(~basic_kdfs::kdf<T>(v)); }
};

// Tagged member KDF: "order the records by this scalar field", i.e. [](const T &e) { return kdf(e.field); } (the
// shapes of radix_tests.cpp:41-43,:111-113; with Descending, README.md:564-574) written as data.  It is an ordinary
// KeyFunc -- calling it gives basic_kdfs::kdf of the field, or its complement -- and the wrapper recognises it: key
// extraction, stable rank sort and the gather of the records then all run on the device (rsx_sort_records_tagged)
// instead of evaluating a lambda once per element on the host.
//     radix_sort(src, aux, n, rsx_kdf::by_member<&sortrec::key>{});
//     radix_sort(src, aux, n, rsx_kdf::by_member<&sortrec::score, true>{});      // descending
template <auto Member, bool Descending = false> struct by_member;
template <typename T, typename F, F T::*Member, bool Descending> struct by_member<Member, Descending> {
	using record_type = T;
	using field_type = F;
	static constexpr bool descending = Descending;
	auto operator()(const T &e) const
	{
		using K = decltype(basic_kdfs::kdf<F>(e.*Member));
		return Descending ? static_cast<K>(~basic_kdfs::kdf<F>(e.*Member)) : basic_kdfs::kdf<F>(e.*Member);
	}
	static size_t offset(const T &e) { return (size_t)(reinterpret_cast<const char *>(&(e.*Member)) - reinterpret_cast<const char *>(&e)); }
};
}  // namespace rsx_kdf

namespace rsx_detail {

template <typename T> constexpr rsx_dtype dtype_of()
{
	if constexpr (std::is_same_v<T, float>) return RSX_F32;
	else if constexpr (std::is_same_v<T, double>) return RSX_F64;
	else if constexpr (std::is_signed_v<T>)
		return sizeof(T) == 1 ? RSX_I8 : sizeof(T) == 2 ? RSX_I16 : sizeof(T) == 4 ? RSX_I32 : RSX_I64;
	else
		return sizeof(T) == 1 ? RSX_U8 : sizeof(T) == 2 ? RSX_U16 : sizeof(T) == 4 ? RSX_U32 : RSX_U64;
}

// Is KeyFunc the default basic_kdfs::kdf<T> / the tagged descending KDF?  (Specialised on "T is a key
// scalar" so that basic_kdfs::kdf<T> is never named for record types.)
// A KeyFunc that is a plain function -- reference or pointer -- only has the default's TYPE: any
// `uint32_t f(const uint32_t&)` has it (the reference takes every callable, radix_sort.hpp:98-99), so for those
// "may_be_default" is decided at run time by the function's address (is_default_kdf below).
template <typename T, typename KeyFunc, bool = basic_kdfs::detail::is_key_scalar_v<T>> struct kdf_kind {
	static constexpr bool may_be_default = false, is_descending = false;
	static bool is_default(const KeyFunc &) { return false; }
};
template <typename T, typename KeyFunc> struct kdf_kind<T, KeyFunc, true> {
	using F = std::remove_cv_t<std::remove_reference_t<KeyFunc>>;
	using D = decltype(basic_kdfs::kdf<T>);                         // the default's function type
	static constexpr bool is_function = std::is_same_v<F, D>, is_pointer = std::is_same_v<F, D *>;
	static constexpr bool may_be_default = is_function || is_pointer;
	static constexpr bool is_descending = std::is_same_v<F, rsx_kdf::descending<T>>;
	template <typename G> static bool is_default(G &kf)
	{
		if constexpr (is_function)
			return &kf == &basic_kdfs::kdf<T>;
		else if constexpr (is_pointer)
			return kf == &basic_kdfs::kdf<T>;
		else
			return false;
	}
};
template <typename T, typename KeyFunc> constexpr bool may_be_default_kdf_v = kdf_kind<T, KeyFunc>::may_be_default;
template <typename T, typename KeyFunc> constexpr bool is_descending_kdf_v = kdf_kind<T, KeyFunc>::is_descending;

template <typename KeyFunc> struct is_member_kdf : std::false_type {};
template <auto M, bool D> struct is_member_kdf<rsx_kdf::by_member<M, D>> : std::true_type {};
template <typename KeyFunc> constexpr bool is_member_kdf_v = is_member_kdf<std::remove_cv_t<std::remove_reference_t<KeyFunc>>>::value;

[[noreturn]] inline void fail(const char *what, int rc)
{
	throw std::runtime_error(std::string(what) + ": rsx error " + std::to_string(rc) + ": " + rsx_last_error());
}

// What rs_sort_main leaves in the caller's (pre-zeroed) Hist, from the counts of loop 1 and what the sort decided:
// nothing for n < 2 (radix_sort.hpp:37-38); the counts after the pre-sorted exit (:48-62) and in skipped columns
// (:64-70); in every kept column the offsets after the exclusive scan (:72-80) and the scatter loop's post-increments
// (:85) -- each bin ends at the end of its run -- all in Hist::value_type arithmetic.
template <typename Hist>
void hist_post_state(Hist &histogram, const uint64_t *counts, const rsx_info &info, size_t key_bytes)
{
	using HVT = typename Hist::value_type;
	bool kept[8] = {false, false, false, false, false, false, false, false};
	if (!info.early_exit)
		for (uint32_t i = 0; i < info.ncols && i < 8; ++i)
			kept[info.cols[i]] = true;
	for (size_t j = 0; j < key_bytes; ++j) {
		if (!kept[j]) {
			for (size_t d = 0; d < 256; ++d)
				histogram[256 * j + d] = static_cast<HVT>(histogram[256 * j + d] + static_cast<HVT>(counts[256 * j + d]));
		} else {
			HVT a = 0;
			for (size_t d = 0; d < 256; ++d) {
				a = static_cast<HVT>(a + static_cast<HVT>(histogram[256 * j + d] + static_cast<HVT>(counts[256 * j + d])));
				histogram[256 * j + d] = a;
			}
		}
	}
}

// counts of loop 1 wanted (a caller-supplied Hist): armed before the sort, disarmed whatever happens
struct hist_capture {
	std::vector<uint64_t> counts;
	explicit hist_capture(bool wanted, size_t key_bytes)
	{
		if (wanted) {
			counts.assign(256 * key_bytes, 0);
			rsx_capture_histogram(counts.data(), counts.size());
		}
	}
	~hist_capture() { rsx_capture_histogram(nullptr, 0); }
};

// KeyType: what the reference derives from the KDF's return type and lets the caller override (the fourth template
// parameter of rs_sort_main, radix_sort.hpp:31): the keys are kf(element) converted to it, its size is the number of columns.
template <typename T, typename KeyFunc>
using derived_key_t = std::remove_cv_t<std::remove_reference_t<std::invoke_result_t<KeyFunc &, const T &>>>;

template <typename T, typename KeyFunc, typename Hist = void, typename KeyType = derived_key_t<T, KeyFunc>>
T *sort_dispatch(T *src, T *aux, size_t n, KeyFunc &&kf, Hist *histogram = nullptr)
{
	static_assert(sizeof(KeyType) <= 8, "KeyType must be 64-bits or less");        // reference radix_sort.hpp:34
	static_assert(std::is_unsigned_v<KeyType>, "KeyType must be unsigned");         // reference radix_sort.hpp:35
	if (n < 2)
		return src;                                                                // reference radix_sort.hpp:37-38
	void *result = nullptr;
	rsx_info info;
	int rc;
	hist_capture cap(histogram != nullptr, sizeof(KeyType));
	// the opaque-callable path: kf once per element on the host, rank sort + gather of the elements on the device
	auto opaque = [&]() {
		std::vector<KeyType> keys(n);
		for (size_t i = 0; i < n; ++i)
			keys[i] = static_cast<KeyType>(kf(src[i]));
		if constexpr (std::is_trivially_copyable_v<T>) {
			return rsx_sort_records(src, aux, n, sizeof(T), keys.data(), sizeof(KeyType), &result, &info);
		} else {
			// Elements with constructors / ownership (the reference move-assigns them, radix_sort.hpp:85-87): the device
			// gives the stable order, the host moves.  aux[i] = move(src[rank[i]]) is the reference's last pass with the
			// earlier ones folded in; an even number of kept columns ends in src (:92), so the elements move back.
			void *ranks = nullptr;
			int r;
			if (n <= 0xFFFFFFFFull) {
				std::vector<uint32_t> ib(2 * n);
				r = rsx_sort_rank_keys(keys.data(), sizeof(KeyType), ib.data(), n, 4, &ranks, &info);
				if (r == RSX_OK && !info.early_exit) {
					const uint32_t *rk = static_cast<const uint32_t *>(ranks);
					for (size_t i = 0; i < n; ++i)
						aux[i] = std::move(src[rk[i]]);
				}
			} else {
				std::vector<uint64_t> ib(2 * n);
				r = rsx_sort_rank_keys(keys.data(), sizeof(KeyType), ib.data(), n, 8, &ranks, &info);
				if (r == RSX_OK && !info.early_exit) {
					const uint64_t *rk = static_cast<const uint64_t *>(ranks);
					for (size_t i = 0; i < n; ++i)
						aux[i] = std::move(src[rk[i]]);
				}
			}
			if (r != RSX_OK)
				return r;
			if (info.early_exit) {                  // pre-sorted: src, aux untouched (radix_sort.hpp:60-62)
				result = src;
			} else if (info.ncols & 1) {
				result = aux;
			} else {
				for (size_t i = 0; i < n; ++i)
					src[i] = std::move(aux[i]);
				result = src;
			}
			info.result_in_aux = result == static_cast<void *>(aux);
			return (int)RSX_OK;
		}
	};
	constexpr bool key_overridden = !std::is_same_v<KeyType, derived_key_t<T, KeyFunc>>;
	if constexpr (key_overridden) {
		rc = opaque();   // (an explicit KeyType: the keys are the KDF's values converted to it, evaluated on the host)
	} else if constexpr (may_be_default_kdf_v<T, KeyFunc>) {
		if (kdf_kind<T, KeyFunc>::is_default(kf))
			rc = rsx_sort(src, aux, n, dtype_of<T>(), RSX_ASCENDING, &result, &info);
		else
			rc = opaque();   // some other function with the default's signature (e.g. `uint32_t desc(const uint32_t &v) { return ~v; }`)
	} else if constexpr (is_descending_kdf_v<T, KeyFunc>) {
		rc = rsx_sort(src, aux, n, dtype_of<T>(), RSX_DESCENDING, &result, &info);
	} else if constexpr (is_member_kdf_v<KeyFunc>) {
		using MK = std::remove_cv_t<std::remove_reference_t<KeyFunc>>;
		static_assert(std::is_trivially_copyable_v<T>, "the GPU path moves elements as raw bytes");
		static_assert(std::is_same_v<typename MK::record_type, T>, "by_member names a field of another type");
		rc = rsx_sort_records_tagged(src, aux, n, sizeof(T), MK::offset(src[0]), dtype_of<typename MK::field_type>(),
		                             MK::descending ? RSX_DESCENDING : RSX_ASCENDING, &result, &info);
	} else {
		rc = opaque();
	}
	if (rc != RSX_OK)
		fail("radix_sort", rc);
	if constexpr (!std::is_void_v<Hist>)
		if (histogram)
			hist_post_state(*histogram, cap.counts.data(), info, sizeof(KeyType));
	return static_cast<T *>(result);
}

}  // namespace rsx_detail

// `passes` is kept for source compatibility; the number of 8-bit columns follows from KeyType.
template <typename T, typename KeyFunc = decltype(basic_kdfs::kdf<T>),
          int passes = sizeof(std::invoke_result_t<KeyFunc &, const T &>)>
T *radix_sort(T *RESTRICT src, T *RESTRICT aux, size_t n, KeyFunc &&kf = basic_kdfs::kdf<T>)
{
	return rsx_detail::sort_dispatch<T>(src, aux, n, kf);
}

// Not in the reference: the same call with the work spread over several MI355X of this process (rsx_sort_multi).
// `devices` holds HIP device indices, one per rank.  Scalar T with the default or the descending KDF.
template <typename T, typename KeyFunc = decltype(basic_kdfs::kdf<T>)>
T *radix_sort_multi(T *RESTRICT src, T *RESTRICT aux, size_t n, const int *devices, int ndev, KeyFunc &&kf = basic_kdfs::kdf<T>)
{
	static_assert(rsx_detail::may_be_default_kdf_v<T, KeyFunc> || rsx_detail::is_descending_kdf_v<T, KeyFunc>,
	              "radix_sort_multi takes scalar keys with basic_kdfs::kdf or rsx_kdf::descending");
	if constexpr (rsx_detail::may_be_default_kdf_v<T, KeyFunc>)
		if (!rsx_detail::kdf_kind<T, KeyFunc>::is_default(kf))
			throw std::invalid_argument("radix_sort_multi: a function other than basic_kdfs::kdf<T> was passed as KeyFunc");
	void *result = nullptr;
	const int rc = rsx_sort_multi(src, aux, n, rsx_detail::dtype_of<T>(),
	                              rsx_detail::is_descending_kdf_v<T, KeyFunc> ? RSX_DESCENDING : RSX_ASCENDING, devices, ndev,
	                              &result, nullptr);
	if (rc != RSX_OK)
		rsx_detail::fail("radix_sort_multi", rc);
	return static_cast<T *>(result);
}

// The reference lets the caller supply the histogram storage (any container with value_type and
// operator[], pre-zeroed, 256 * sizeof(KeyType) entries: radix_sort.hpp:28-33).  The device keeps its own counters;
// the counts of its histogram pass are brought back (rsx_capture_histogram) and `histogram` is left in the state the
// reference leaves it in (rsx_detail::hist_post_state; pinned against the real rs_sort_main by tests/golden, hist_post).
// The fourth template parameter is the reference's (radix_sort.hpp:31): KeyType, by default the KDF's return type.
template <typename T, typename KeyFunc = decltype(basic_kdfs::kdf<T>), typename Hist,
          typename KeyType = rsx_detail::derived_key_t<T, KeyFunc>>
T *rs_sort_main(T *RESTRICT src, T *RESTRICT aux, size_t n, Hist &histogram, KeyFunc &&kf = basic_kdfs::kdf<T>)
{
	return rsx_detail::sort_dispatch<T, KeyFunc, Hist, KeyType>(src, aux, n, std::forward<KeyFunc>(kf), &histogram);
}
