/*
 * ref_driver.cpp -- C-ABI doorway into the REAL reference templates.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is ours; the headers it includes are
 * compiled from where they lie (-I/root/reference, see oracle/Makefile) and are
 * never copied into this repository.  The resulting oracle/_ref/libref.so is
 * git-ignored; it travels to the GPU box with the tree and is used there as
 * (a) a second checker next to oracle/rs_oracle.c and (b) bench.py's
 * cpu_baseline of kind "reference".
 *
 * Instantiates:
 *   radix_sort<T>(src, aux, n)                 radix_sort.hpp:98-115
 *   radix_sort(src, aux, n, ~kdf)              README.md:564-574 (descending)
 *   radix_sort on {key, payload} records       radix_tests.cpp:41-43 shape
 *   radix_sort_rank<T, IdxType>(...)           radix_sort_rank.hpp:97-112
 *   rs_sort_main(src, aux, n, Hist&)           radix_sort.hpp:31-93 (caller-supplied histogram)
 */
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <type_traits>
#include <vector>

#include "radix_sort.hpp"
#include "radix_sort_rank.hpp"

namespace {

template <typename T>
int sort_scalar(void *src_v, void *aux_v, size_t n, int order)
{
	T *src = static_cast<T *>(src_v);
	T *aux = static_cast<T *>(aux_v);
	T *res;
	if (order == 0) {
		res = radix_sort(src, aux, n);
	} else {
		using KT = decltype(basic_kdfs::kdf(std::declval<const T &>()));
		auto desc = [](const T &v) -> KT { return static_cast<KT>(~basic_kdfs::kdf(v)); };
		res = radix_sort(src, aux, n, desc);
	}
	return res == src ? 0 : 1;
}

template <typename T, typename Idx>
int rank_scalar_idx(const void *src_v, void *ib_v, size_t n, int order)
{
	const T *src = static_cast<const T *>(src_v);
	Idx *ib = static_cast<Idx *>(ib_v);
	Idx *res;
	if (order == 0) {
		res = radix_sort_rank(src, ib, n);
	} else {
		using KT = decltype(basic_kdfs::kdf(std::declval<const T &>()));
		auto desc = [](const T &v) -> KT { return static_cast<KT>(~basic_kdfs::kdf(v)); };
		res = radix_sort_rank(src, ib, n, desc);
	}
	return res == ib ? 0 : 1;
}

template <typename T>
int rank_scalar(const void *src, void *ib, size_t n, int idx_bytes, int order)
{
	switch (idx_bytes) {
	case 1: return rank_scalar_idx<T, uint8_t>(src, ib, n, order);
	case 2: return rank_scalar_idx<T, uint16_t>(src, ib, n, order);
	case 4: return rank_scalar_idx<T, uint32_t>(src, ib, n, order);
	case 8: return rank_scalar_idx<T, uint64_t>(src, ib, n, order);
	default: return -1;
	}
}

template <typename K, typename V>
struct KV {
	K k;
	V v;
};

template <typename K, typename V>
int sort_kv(void *src_v, void *aux_v, size_t n, int order)
{
	using R = KV<K, V>;
	using KT = decltype(basic_kdfs::kdf(std::declval<const K &>()));
	R *src = static_cast<R *>(src_v);
	R *aux = static_cast<R *>(aux_v);
	R *res;
	if (order == 0) {
		auto kf = [](const R &e) -> KT { return basic_kdfs::kdf(e.k); };
		res = radix_sort(src, aux, n, kf);
	} else {
		auto kf = [](const R &e) -> KT { return static_cast<KT>(~basic_kdfs::kdf(e.k)); };
		res = radix_sort(src, aux, n, kf);
	}
	return res == src ? 0 : 1;
}

/* rs_sort_main with a caller-supplied Hist (radix_sort.hpp:28-33): std::vector<HVT>, pre-zeroed */
template <typename T, typename HVT>
int sort_main_hist_hvt(void *src_v, void *aux_v, size_t n, uint64_t *hist_out)
{
	using KT = decltype(basic_kdfs::kdf(std::declval<const T &>()));
	std::vector<HVT> hist(256 * sizeof(KT), 0);
	T *src = static_cast<T *>(src_v);
	T *aux = static_cast<T *>(aux_v);
	T *res = rs_sort_main(src, aux, n, hist);
	for (size_t i = 0; i < hist.size(); ++i)
		hist_out[i] = hist[i];
	return res == src ? 0 : 1;
}

template <typename T>
int sort_main_hist(void *src, void *aux, size_t n, int hvt_bytes, uint64_t *hist_out)
{
	switch (hvt_bytes) {
	case 1: return sort_main_hist_hvt<T, uint8_t>(src, aux, n, hist_out);
	case 2: return sort_main_hist_hvt<T, uint16_t>(src, aux, n, hist_out);
	case 4: return sort_main_hist_hvt<T, uint32_t>(src, aux, n, hist_out);
	case 8: return sort_main_hist_hvt<T, uint64_t>(src, aux, n, hist_out);
	default: return -1;
	}
}

/* radix_tests.cpp:15-18 record shape: 1-byte key, pointer-sized payload */
struct sortrec {
	uint8_t key;
	const char *name;
};

} // namespace

/* dtype codes are oracle/rs_oracle.h's RSO_* */
extern "C" {

int ref_sort(void *src, void *aux, size_t n, int dtype, int order)
{
	switch (dtype) {
	case 0: return sort_scalar<uint8_t>(src, aux, n, order);
	case 1: return sort_scalar<uint16_t>(src, aux, n, order);
	case 2: return sort_scalar<uint32_t>(src, aux, n, order);
	case 3: return sort_scalar<uint64_t>(src, aux, n, order);
	case 4: return sort_scalar<int8_t>(src, aux, n, order);
	case 5: return sort_scalar<int16_t>(src, aux, n, order);
	case 6: return sort_scalar<int32_t>(src, aux, n, order);
	case 7: return sort_scalar<int64_t>(src, aux, n, order);
	case 8: return sort_scalar<float>(src, aux, n, order);
	case 9: return sort_scalar<double>(src, aux, n, order);
	default: return -1;
	}
}

int ref_sort_rank(const void *src, void *index_buffer, size_t n, int dtype, int idx_bytes, int order)
{
	switch (dtype) {
	case 0: return rank_scalar<uint8_t>(src, index_buffer, n, idx_bytes, order);
	case 1: return rank_scalar<uint16_t>(src, index_buffer, n, idx_bytes, order);
	case 2: return rank_scalar<uint32_t>(src, index_buffer, n, idx_bytes, order);
	case 3: return rank_scalar<uint64_t>(src, index_buffer, n, idx_bytes, order);
	case 4: return rank_scalar<int8_t>(src, index_buffer, n, idx_bytes, order);
	case 5: return rank_scalar<int16_t>(src, index_buffer, n, idx_bytes, order);
	case 6: return rank_scalar<int32_t>(src, index_buffer, n, idx_bytes, order);
	case 7: return rank_scalar<int64_t>(src, index_buffer, n, idx_bytes, order);
	case 8: return rank_scalar<float>(src, index_buffer, n, idx_bytes, order);
	case 9: return rank_scalar<double>(src, index_buffer, n, idx_bytes, order);
	default: return -1;
	}
}

/* rs_sort_main(src, aux, n, hist) -- radix_sort.hpp:31-93 -- with a zeroed std::vector<HVT> of
 * hvt_bytes-wide counters; hist_out (256 * key bytes uint64) receives what it holds on return. */
int ref_sort_main_hist(void *src, void *aux, size_t n, int dtype, int hvt_bytes, uint64_t *hist_out)
{
	switch (dtype) {
	case 0: return sort_main_hist<uint8_t>(src, aux, n, hvt_bytes, hist_out);
	case 1: return sort_main_hist<uint16_t>(src, aux, n, hvt_bytes, hist_out);
	case 2: return sort_main_hist<uint32_t>(src, aux, n, hvt_bytes, hist_out);
	case 3: return sort_main_hist<uint64_t>(src, aux, n, hvt_bytes, hist_out);
	case 4: return sort_main_hist<int8_t>(src, aux, n, hvt_bytes, hist_out);
	case 5: return sort_main_hist<int16_t>(src, aux, n, hvt_bytes, hist_out);
	case 6: return sort_main_hist<int32_t>(src, aux, n, hvt_bytes, hist_out);
	case 7: return sort_main_hist<int64_t>(src, aux, n, hvt_bytes, hist_out);
	case 8: return sort_main_hist<float>(src, aux, n, hvt_bytes, hist_out);
	case 9: return sort_main_hist<double>(src, aux, n, hvt_bytes, hist_out);
	default: return -1;
	}
}

/* {key; payload} records with padding-free layouts only */
int ref_sort_kv(void *src, void *aux, size_t n, int key_dtype, int payload_bytes, int order)
{
	if (payload_bytes == 4) {
		switch (key_dtype) {
		case 2: return sort_kv<uint32_t, uint32_t>(src, aux, n, order);
		case 6: return sort_kv<int32_t, uint32_t>(src, aux, n, order);
		case 8: return sort_kv<float, uint32_t>(src, aux, n, order);
		default: return -1;
		}
	}
	if (payload_bytes == 8) {
		switch (key_dtype) {
		case 3: return sort_kv<uint64_t, uint64_t>(src, aux, n, order);
		case 7: return sort_kv<int64_t, uint64_t>(src, aux, n, order);
		case 9: return sort_kv<double, uint64_t>(src, aux, n, order);
		default: return -1;
		}
	}
	return -1;
}

/* radix_tests.cpp:45-69 (order 0, kdf = entry.key) and the complemented form */
int ref_sort_sortrec(void *src_v, void *aux_v, size_t n, int order)
{
	sortrec *src = static_cast<sortrec *>(src_v);
	sortrec *aux = static_cast<sortrec *>(aux_v);
	sortrec *res;
	if (order == 0) {
		auto kf = [](const sortrec &e) -> uint8_t { return e.key; };
		res = radix_sort(src, aux, n, kf);
	} else {
		auto kf = [](const sortrec &e) -> uint8_t { return static_cast<uint8_t>(~e.key); };
		res = radix_sort(src, aux, n, kf);
	}
	return res == src ? 0 : 1;
}

/* radix_tests.cpp:71-105: rank sort of the records, IdxType = uint8_t */
int ref_rank_sortrec_u8idx(const void *src_v, void *ib_v, size_t n)
{
	const sortrec *src = static_cast<const sortrec *>(src_v);
	uint8_t *ib = static_cast<uint8_t *>(ib_v);
	auto kf = [](const sortrec &e) -> uint8_t { return e.key; };
	uint8_t *res = radix_sort_rank(src, ib, n, kf);
	return res == ib ? 0 : 1;
}

size_t ref_sizeof_sortrec(void) { return sizeof(sortrec); }

} // extern "C"
