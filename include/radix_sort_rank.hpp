// radix_sort_rank.hpp -- drop-in for the reference's rank (argsort) header, backed by librsx.so.
//
//     IdxType* radix_sort_rank(const T* src, IdxType* index_buffer /* 2n entries */, size_t n, KeyFunc&& kf)
//     IdxType* rs_sort_rank(src, index_buffer, n, Hist& histogram, kf)          (reference :97-98, :22-23)
//
// Returns the half of index_buffer that holds the stable ranks: the first half when the number of
// non-constant key columns is even (also for n < 2 and pre-sorted input, where it is 0..n-1), the
// second half (index_buffer + n) when it is odd.  The ranks are those of a correct stable argsort,
// i.e. of Listing 6 (radix_sort_u32_ranks.c); the reference header's own loop reads src[j] instead of
// src[index[j]] on later passes and is only right for single-column keys.
#pragma once

#include <cstddef>
#include <cstdint>
#include <type_traits>
#include <vector>

#include "radix_sort.hpp"

namespace rsx_detail {

template <typename T, typename IdxType, typename KeyFunc, typename Hist = void>
IdxType *rank_dispatch(const T *src, IdxType *index_buffer, size_t n, KeyFunc &&kf, Hist *histogram = nullptr)
{
	using KeyType = std::remove_cv_t<std::remove_reference_t<std::invoke_result_t<KeyFunc &, const T &>>>;
	static_assert(sizeof(KeyType) <= 8, "KeyType must be 64-bits or less");
	static_assert(std::is_unsigned_v<KeyType>, "KeyType must be unsigned");
	static_assert(std::is_integral_v<IdxType> && (sizeof(IdxType) == 1 || sizeof(IdxType) == 2 || sizeof(IdxType) == 4 ||
	                                             sizeof(IdxType) == 8), "IdxType must be a 1/2/4/8-byte integer");
	if (n < 2) {                                                                  // reference radix_sort_rank.hpp:28-32
		if (n != 0)
			index_buffer[0] = 0;
		return index_buffer;
	}
	void *result = nullptr;
	rsx_info info;
	int rc;
	hist_capture cap(histogram != nullptr, sizeof(KeyType));
	auto opaque = [&]() {   // kf once per element on the host, the rank sort of those keys on the device
		std::vector<KeyType> keys(n);
		for (size_t i = 0; i < n; ++i)
			keys[i] = kf(src[i]);
		return rsx_sort_rank_keys(keys.data(), sizeof(KeyType), index_buffer, n, sizeof(IdxType), &result, &info);
	};
	if constexpr (may_be_default_kdf_v<T, KeyFunc>) {
		if (kdf_kind<T, KeyFunc>::is_default(kf))   // (a plain function: by address, see radix_sort.hpp)
			rc = rsx_sort_rank(src, index_buffer, n, dtype_of<T>(), sizeof(IdxType), RSX_ASCENDING, &result, &info);
		else
			rc = opaque();
	} else if constexpr (is_descending_kdf_v<T, KeyFunc>) {
		rc = rsx_sort_rank(src, index_buffer, n, dtype_of<T>(), sizeof(IdxType), RSX_DESCENDING, &result, &info);
	} else {
		rc = opaque();
	}
	if (rc != RSX_OK)
		fail("radix_sort_rank", rc);
	if constexpr (!std::is_void_v<Hist>)
		if (histogram)
			hist_post_state(*histogram, cap.counts.data(), info, sizeof(KeyType));   // radix_sort_rank.hpp:44-88: as rs_sort_main
	return static_cast<IdxType *>(result);
}

}  // namespace rsx_detail

template <typename T, typename IdxType, typename KeyFunc = decltype(basic_kdfs::kdf<T>),
          int passes = sizeof(std::invoke_result_t<KeyFunc &, const T &>)>
IdxType *radix_sort_rank(const T *RESTRICT src, IdxType *RESTRICT index_buffer, size_t n, KeyFunc &&kf = basic_kdfs::kdf<T>)
{
	return rsx_detail::rank_dispatch<T, IdxType>(src, index_buffer, n, kf);
}

template <typename T, typename KeyFunc = decltype(basic_kdfs::kdf<T>), typename Hist, typename IdxType = size_t>
IdxType *rs_sort_rank(const T *RESTRICT src, IdxType *RESTRICT index_buffer, size_t n, Hist &histogram,
                      KeyFunc &&kf = basic_kdfs::kdf<T>)
{
	return rsx_detail::rank_dispatch<T, IdxType>(src, index_buffer, n, kf, &histogram);
}
