/*
 * rs_oracle.c -- CPU restatement of eloj/radix-sorting's LSD radix sort.
 *
 * TEST INFRASTRUCTURE ONLY (see rs_oracle.h).  Parity status: PINNED against
 * the reference's known-answer outputs, tests/golden/kat_table.json and
 * oracle/_ref/libref.so (the real reference headers compiled in place).
 *
 * The algorithm restated here (citations relative to the reference repo):
 *   radix_sort.hpp:31-93    rs_sort_main: histogram + pre-sort detect (:48-58),
 *                           early exit (:60-62), column-skip probe (:64-70),
 *                           exclusive scan (:72-80), scatter passes (:82-90),
 *                           returned pointer (:92)
 *   radix_sort.hpp:98-115   radix_sort: n<2 exit, counter-width choice
 *   radix_sort_rank.hpp:22-92, radix_sort_u32_ranks.c:85-107   rank variant
 *   radix_sort_basic_kdf.hpp:13-46   key derivation
 */
#include "rs_oracle.h"

#include <stdlib.h>
#include <string.h>

/* ---------------------------------------------------------------- KDFs --- */

size_t rso_dtype_size(int dtype)
{
	switch (dtype) {
	case RSO_U8:  case RSO_I8:  return 1;
	case RSO_U16: case RSO_I16: return 2;
	case RSO_U32: case RSO_I32: case RSO_F32: return 4;
	case RSO_U64: case RSO_I64: case RSO_F64: return 8;
	default: return 0;
	}
}

/* radix_sort_basic_kdf.hpp:19-23 (unsigned: identity), :26-30 (signed: flip the
 * top bit, highbit<T> :13-17), :32-38 (float), :40-46 (double).  The result is
 * sizeof(T) bytes wide; `order` complements it within that width
 * (README.md:564-574).  Returned zero-extended. */
static inline uint64_t kdf_bits(uint64_t raw, int dtype, int order)
{
	uint64_t k = raw;
	unsigned bits = (unsigned)rso_dtype_size(dtype) * 8u;
	uint64_t width_mask = bits == 64 ? ~0ULL : ((1ULL << bits) - 1ULL);
	switch (dtype) {
	case RSO_I8: case RSO_I16: case RSO_I32: case RSO_I64:
		k = raw ^ (1ULL << (bits - 1));
		break;
	case RSO_F32: {
		uint32_t u = (uint32_t)raw;
		k = (uint32_t)(u ^ ((uint32_t)(-(int32_t)(u >> 31)) | (1UL << 31)));
		break;
	}
	case RSO_F64:
		k = raw ^ ((uint64_t)(-(int64_t)(raw >> 63)) | (1ULL << 63));
		break;
	default:
		break;
	}
	if (order == RSO_DESCENDING)
		k = ~k;
	return k & width_mask;
}

static inline uint64_t load_raw(const void *p, size_t bytes)
{
	uint64_t v = 0;
	memcpy(&v, p, bytes); /* little-endian host, as the reference assumes */
	return v;
}

uint64_t rso_kdf(const void *elem, int dtype, int order)
{
	return kdf_bits(load_raw(elem, rso_dtype_size(dtype)), dtype, order);
}

/* ------------------------------------------------- generic record sort --- */

/* radix_sort.hpp:48-58 */
void rso_histogram(const void *src, size_t n, size_t rec_size, size_t key_off,
                   int dtype, int order, uint64_t *hist, uint64_t *n_unsorted)
{
	const unsigned char *s = (const unsigned char *)src;
	const size_t kb = rso_dtype_size(dtype);
	uint64_t unsorted = n;
	memset(hist, 0, sizeof(uint64_t) * 256 * 8);
	for (size_t i = 0; i < n; ++i) {
		uint64_t key0 = kdf_bits(load_raw(s + i * rec_size + key_off, kb), dtype, order);
		if (i < n - 1) {
			uint64_t key1 = kdf_bits(load_raw(s + (i + 1) * rec_size + key_off, kb), dtype, order);
			if (key0 <= key1)
				--unsorted;
		}
		for (size_t j = 0; j < kb; ++j)
			++hist[256 * j + ((key0 >> (8 * j)) & 0xFF)];
	}
	if (n_unsorted)
		*n_unsorted = unsorted;
}

static void info_reset(rso_info *info, int dtype)
{
	if (!info)
		return;
	memset(info, 0, sizeof(*info));
	info->key_bytes = (uint32_t)rso_dtype_size(dtype);
}

/* Shared front half of rs_sort_main / rs_sort_rank: loop 1, the early exit,
 * the column probe and the exclusive scan.  Returns the number of kept
 * columns, or -1 for the pre-sorted early exit. */
static int plan_passes(const void *src, size_t n, size_t rec_size, size_t key_off,
                       int dtype, int order, uint64_t *hist, unsigned *cols, rso_info *info)
{
	const size_t kb = rso_dtype_size(dtype);
	uint64_t n_unsorted;
	rso_histogram(src, n, rec_size, key_off, dtype, order, hist, &n_unsorted);
	if (info)
		info->n_unsorted = n_unsorted;
	if (n_unsorted < 2) /* radix_sort.hpp:60-62 */
		return -1;

	/* radix_sort.hpp:64-70: probe with the first key only */
	uint64_t key0 = kdf_bits(load_raw((const unsigned char *)src + key_off, kb), dtype, order);
	int ncols = 0;
	for (size_t i = 0; i < kb; ++i)
		if (hist[256 * i + ((key0 >> (8 * i)) & 0xFF)] != n)
			cols[ncols++] = (unsigned)i;

	/* radix_sort.hpp:72-80 */
	for (int i = 0; i < ncols; ++i) {
		uint64_t a = 0;
		for (unsigned j = 0; j < 256; ++j) {
			uint64_t b = hist[256 * cols[i] + j];
			hist[256 * cols[i] + j] = a;
			a += b;
		}
	}
	if (info) {
		info->ncols = (uint32_t)ncols;
		for (int i = 0; i < ncols; ++i)
			info->cols[i] = cols[i];
	}
	return ncols;
}

/* hist_out (may be NULL; 256 * key bytes entries): the state rs_sort_main leaves in the caller's
 * pre-zeroed Hist (radix_sort.hpp:28-33): untouched for n < 2 (:37-38), the raw counts after the
 * pre-sorted exit (:48-62) and in skipped columns, and in every kept column the offsets after the
 * exclusive scan (:72-80) and the post-increments of the scatter loop (:85) -- i.e. end offsets. */
static int sort_records_impl(void *src_v, void *aux_v, size_t n, size_t rec_size, size_t key_off,
                             int dtype, int order, rso_info *info, uint64_t *hist_out)
{
	unsigned char *src = (unsigned char *)src_v, *aux = (unsigned char *)aux_v;
	const size_t kb = rso_dtype_size(dtype);
	uint64_t *hist;
	unsigned cols[8];
	int swapped = 0;

	info_reset(info, dtype);
	if (n < 2) { /* radix_sort.hpp:37-38, :100-101 */
		if (info)
			info->early_exit = 1;
		return 0;
	}
	hist = (uint64_t *)malloc(sizeof(uint64_t) * 256 * 8);
	int ncols = plan_passes(src, n, rec_size, key_off, dtype, order, hist, cols, info);
	if (ncols < 0) {
		if (info)
			info->early_exit = 2;
		if (hist_out)
			memcpy(hist_out, hist, sizeof(uint64_t) * 256 * kb);
		free(hist);
		return 0;
	}

	/* radix_sort.hpp:82-90: in-order traversal, post-increment => stable */
	for (int i = 0; i < ncols; ++i) {
		uint64_t *h = hist + 256 * cols[i];
		unsigned shift = 8 * cols[i];
		for (size_t j = 0; j < n; ++j) {
			const unsigned char *k = src + j * rec_size;
			uint64_t key = kdf_bits(load_raw(k + key_off, kb), dtype, order);
			size_t dst = h[(key >> shift) & 0xFF]++;
			memcpy(aux + dst * rec_size, k, rec_size);
		}
		unsigned char *t = src; src = aux; aux = t;
		swapped ^= 1;
	}
	if (hist_out)
		memcpy(hist_out, hist, sizeof(uint64_t) * 256 * kb);
	free(hist);
	if (info)
		info->result_in_aux = (uint32_t)swapped;
	return swapped; /* radix_sort.hpp:92 */
}

int rso_sort_records(void *src_v, void *aux_v, size_t n, size_t rec_size, size_t key_off,
                     int dtype, int order, rso_info *info)
{
	return sort_records_impl(src_v, aux_v, n, rec_size, key_off, dtype, order, info, NULL);
}

/* rs_sort_main(src, aux, n, histogram, kf) with a caller-supplied Hist whose value_type has
 * hvt_bytes bytes (radix_sort.hpp:28-33; :102-114 picks 1/2/4/8 by n).  hist_out receives the
 * Hist's final contents, each entry reduced modulo 2^(8 hvt_bytes) as HVT arithmetic does. */
int rso_sort_main_hist(void *src, void *aux, size_t n, int dtype, int order, int hvt_bytes,
                       uint64_t *hist_out, rso_info *info)
{
	const size_t kb = rso_dtype_size(dtype);
	memset(hist_out, 0, sizeof(uint64_t) * 256 * kb);
	const int r = sort_records_impl(src, aux, n, kb, 0, dtype, order, info, hist_out);
	if (hvt_bytes < 8)
		for (size_t i = 0; i < 256 * kb; ++i)
			hist_out[i] &= (1ull << (8 * hvt_bytes)) - 1;
	return r;
}

/* ----------------------------------------------- typed scalar fast path --- */

/* One instantiation per scalar type, so that the loops compile to what the
 * reference template compiles to (this is the timed "port" baseline). */
#define RSO_DEFINE_SORT(NAME, T, UT, DTYPE)                                              \
static int NAME(T *src, T *aux, size_t n, int order, rso_info *info)                      \
{                                                                                          \
	enum { WC = sizeof(UT) };                                                              \
	size_t *hist = (size_t *)calloc(256 * WC, sizeof(size_t));                             \
	unsigned cols[8], ncols = 0;                                                           \
	int swapped = 0;                                                                       \
	size_t n_unsorted = n;                                                                 \
	UT key0;                                                                               \
	for (size_t i = 0; i < n; ++i) { /* radix_sort.hpp:49-58 */                            \
		key0 = (UT)kdf_bits((UT)load_raw(src + i, sizeof(T)), DTYPE, order);               \
		if ((i < n - 1) &&                                                                 \
		    (key0 <= (UT)kdf_bits((UT)load_raw(src + i + 1, sizeof(T)), DTYPE, order)))    \
			--n_unsorted;                                                                  \
		for (unsigned j = 0; j < WC; ++j)                                                  \
			++hist[256 * j + ((key0 >> (8 * j)) & 0xFF)];                                  \
	}                                                                                      \
	if (info)                                                                              \
		info->n_unsorted = n_unsorted;                                                     \
	if (n_unsorted < 2) { /* radix_sort.hpp:60-62 */                                       \
		if (info)                                                                          \
			info->early_exit = 2;                                                          \
		free(hist);                                                                        \
		return 0;                                                                          \
	}                                                                                      \
	key0 = (UT)kdf_bits((UT)load_raw(src, sizeof(T)), DTYPE, order); /* :64-70 */          \
	for (unsigned i = 0; i < WC; ++i)                                                      \
		if (hist[256 * i + ((key0 >> (8 * i)) & 0xFF)] != n)                               \
			cols[ncols++] = i;                                                             \
	for (unsigned i = 0; i < ncols; ++i) { /* :72-80 */                                    \
		size_t a = 0;                                                                      \
		for (unsigned j = 0; j < 256; ++j) {                                               \
			size_t b = hist[256 * cols[i] + j];                                            \
			hist[256 * cols[i] + j] = a;                                                   \
			a += b;                                                                        \
		}                                                                                  \
	}                                                                                      \
	for (unsigned i = 0; i < ncols; ++i) { /* :82-90 */                                    \
		size_t *h = hist + 256 * cols[i];                                                  \
		const unsigned shift = 8 * cols[i];                                                \
		for (size_t j = 0; j < n; ++j) {                                                   \
			T k = src[j];                                                                  \
			UT key = (UT)kdf_bits((UT)load_raw(&k, sizeof(T)), DTYPE, order);              \
			size_t dst = h[(key >> shift) & 0xFF]++;                                       \
			aux[dst] = k;                                                                  \
		}                                                                                  \
		T *t = src; src = aux; aux = t;                                                    \
		swapped ^= 1;                                                                      \
	}                                                                                      \
	free(hist);                                                                            \
	if (info) {                                                                            \
		info->ncols = ncols;                                                               \
		for (unsigned i = 0; i < ncols; ++i)                                               \
			info->cols[i] = cols[i];                                                       \
		info->result_in_aux = (uint32_t)swapped;                                           \
	}                                                                                      \
	return swapped; /* :92 */                                                              \
}

RSO_DEFINE_SORT(sort_u8,  uint8_t,  uint8_t,  RSO_U8)
RSO_DEFINE_SORT(sort_u16, uint16_t, uint16_t, RSO_U16)
RSO_DEFINE_SORT(sort_u32, uint32_t, uint32_t, RSO_U32)
RSO_DEFINE_SORT(sort_u64, uint64_t, uint64_t, RSO_U64)
RSO_DEFINE_SORT(sort_i8,  int8_t,   uint8_t,  RSO_I8)
RSO_DEFINE_SORT(sort_i16, int16_t,  uint16_t, RSO_I16)
RSO_DEFINE_SORT(sort_i32, int32_t,  uint32_t, RSO_I32)
RSO_DEFINE_SORT(sort_i64, int64_t,  uint64_t, RSO_I64)
RSO_DEFINE_SORT(sort_f32, float,    uint32_t, RSO_F32)
RSO_DEFINE_SORT(sort_f64, double,   uint64_t, RSO_F64)

int rso_sort(void *src, void *aux, size_t n, int dtype, int order, rso_info *info)
{
	info_reset(info, dtype);
	if (n < 2) { /* radix_sort.hpp:100-101 */
		if (info)
			info->early_exit = 1;
		return 0;
	}
	switch (dtype) {
	case RSO_U8:  return sort_u8((uint8_t *)src, (uint8_t *)aux, n, order, info);
	case RSO_U16: return sort_u16((uint16_t *)src, (uint16_t *)aux, n, order, info);
	case RSO_U32: return sort_u32((uint32_t *)src, (uint32_t *)aux, n, order, info);
	case RSO_U64: return sort_u64((uint64_t *)src, (uint64_t *)aux, n, order, info);
	case RSO_I8:  return sort_i8((int8_t *)src, (int8_t *)aux, n, order, info);
	case RSO_I16: return sort_i16((int16_t *)src, (int16_t *)aux, n, order, info);
	case RSO_I32: return sort_i32((int32_t *)src, (int32_t *)aux, n, order, info);
	case RSO_I64: return sort_i64((int64_t *)src, (int64_t *)aux, n, order, info);
	case RSO_F32: return sort_f32((float *)src, (float *)aux, n, order, info);
	case RSO_F64: return sort_f64((double *)src, (double *)aux, n, order, info);
	default: return -1;
	}
}

/* ------------------------------------------------------------ rank sort --- */

static inline uint64_t idx_load(const unsigned char *p, int idx_bytes)
{
	uint64_t v = 0;
	memcpy(&v, p, (size_t)idx_bytes);
	return v;
}

static inline void idx_store(unsigned char *p, int idx_bytes, uint64_t v)
{
	memcpy(p, &v, (size_t)idx_bytes); /* truncates like the IdxType assignment at radix_sort_rank.hpp:52 */
}

static int sort_rank_impl(const void *src_v, size_t rec_size, size_t key_off, int dtype, int order,
                          void *index_buffer, int idx_bytes, size_t n, rso_info *info,
                          int gather_through_index)
{
	const unsigned char *src = (const unsigned char *)src_v;
	unsigned char *ib = (unsigned char *)index_buffer;
	const size_t kb = rso_dtype_size(dtype);
	unsigned cols[8];

	info_reset(info, dtype);
	if (n < 2) { /* radix_sort_rank.hpp:28-32 */
		if (n != 0)
			idx_store(ib, idx_bytes, 0);
		if (info)
			info->early_exit = 1;
		return 0;
	}
	/* radix_sort_rank.hpp:47-53: loop 1 also writes index_buffer[i] = i */
	for (size_t i = 0; i < n; ++i)
		idx_store(ib + i * (size_t)idx_bytes, idx_bytes, i);

	uint64_t *hist = (uint64_t *)malloc(sizeof(uint64_t) * 256 * 8);
	int ncols = plan_passes(src, n, rec_size, key_off, dtype, order, hist, cols, info);
	if (ncols < 0) { /* radix_sort_rank.hpp:55-57 */
		if (info)
			info->early_exit = 2;
		free(hist);
		return 0;
	}

	unsigned char *isrc = ib;                          /* radix_sort_rank.hpp:77 */
	unsigned char *idst = ib + n * (size_t)idx_bytes;  /* radix_sort_rank.hpp:78 */
	int swapped = 0;
	for (int i = 0; i < ncols; ++i) {
		uint64_t *h = hist + 256 * cols[i];
		unsigned shift = 8 * cols[i];
		for (size_t j = 0; j < n; ++j) {
			uint64_t idx = idx_load(isrc + j * (size_t)idx_bytes, idx_bytes);
			/* Listing 6 (radix_sort_u32_ranks.c:92,98,104) reads the key of the
			 * element the j-th index points at; the header reads src[j]
			 * (radix_sort_rank.hpp:82-83). */
			size_t at = gather_through_index ? (size_t)idx : j;
			uint64_t key = kdf_bits(load_raw(src + at * rec_size + key_off, kb), dtype, order);
			size_t dst = h[(key >> shift) & 0xFF]++;
			idx_store(idst + dst * (size_t)idx_bytes, idx_bytes, idx);
		}
		unsigned char *t = isrc; isrc = idst; idst = t;
		swapped ^= 1;
	}
	free(hist);
	if (info)
		info->result_in_aux = (uint32_t)swapped;
	return swapped; /* radix_sort_rank.hpp:91 */
}

int rso_sort_rank(const void *src, size_t rec_size, size_t key_off, int dtype, int order,
                  void *index_buffer, int idx_bytes, size_t n, rso_info *info)
{
	return sort_rank_impl(src, rec_size, key_off, dtype, order, index_buffer, idx_bytes, n, info, 1);
}

int rso_sort_rank_asheader(const void *src, size_t rec_size, size_t key_off, int dtype, int order,
                           void *index_buffer, int idx_bytes, size_t n, rso_info *info)
{
	return sort_rank_impl(src, rec_size, key_off, dtype, order, index_buffer, idx_bytes, n, info, 0);
}

/* -------------------------------------------------------- test helpers --- */

uint64_t rso_fnv1a64(const void *data, size_t bytes)
{
	const unsigned char *p = (const unsigned char *)data;
	uint64_t h = 0xcbf29ce484222325ULL;
	for (size_t i = 0; i < bytes; ++i) {
		h ^= p[i];
		h *= 0x100000001b3ULL;
	}
	return h;
}

static inline uint64_t splitmix64(uint64_t *s)
{
	uint64_t z = (*s += 0x9E3779B97F4A7C15ULL);
	z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
	z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
	return z ^ (z >> 31);
}

void rso_fill_splitmix(void *dst, size_t n, size_t elem_size, uint64_t seed, uint64_t mask)
{
	unsigned char *d = (unsigned char *)dst;
	uint64_t s = seed;
	for (size_t i = 0; i < n; ++i) {
		uint64_t v = splitmix64(&s) & mask;
		memcpy(d + i * elem_size, &v, elem_size);
	}
}
