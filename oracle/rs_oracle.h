/*
 * rs_oracle.h -- CPU restatement of eloj/radix-sorting's LSD radix sort.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load it, and only as the checker / reported baseline.  The product path
 * (radix_sorting_amd/csrc + include/) never links, loads or calls it.
 *
 * Parity status: PINNED.  The restatement is checked (tests/test_oracle_*.py)
 * against (a) the reference's own known-answer outputs (radix_tests.cpp,
 * Listings 3-6 stdout, README.md:612-623 float order), (b) the hash table in
 * tests/golden/kat_table.json generated from the real reference headers by
 * tools/gen_golden.py, and (c) oracle/_ref/libref.so -- the real reference
 * headers compiled in place from /root/reference -- on randomized sweeps.
 *
 * Every function cites the reference file:line it restates
 * (paths relative to the reference repo root).
 */
#ifndef RS_ORACLE_H
#define RS_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Key kinds: the scalar types radix_experiment.cpp:264-282 dispatches on plus
 * the 8/16-bit ones basic_kdfs::kdf also accepts (radix_sort_basic_kdf.hpp:19-46).
 * Values are shared with include/rsx.h (rsx_dtype). */
enum {
	RSO_U8 = 0, RSO_U16 = 1, RSO_U32 = 2, RSO_U64 = 3,
	RSO_I8 = 4, RSO_I16 = 5, RSO_I32 = 6, RSO_I64 = 7,
	RSO_F32 = 8, RSO_F64 = 9
};

/* Key order: 0 = basic_kdfs::kdf as is; 1 = bitwise complement of it, the
 * "descending" KDF of README.md:564-574 / radix_tests.cpp:111-113,175-177. */
enum { RSO_ASCENDING = 0, RSO_DESCENDING = 1 };

typedef struct rso_info {
	uint64_t n_unsorted;    /* radix_sort.hpp:48-58 counter after loop 1      */
	uint32_t key_bytes;     /* wc = sizeof(KeyType), radix_sort.hpp:40        */
	uint32_t ncols;         /* kept columns, radix_sort.hpp:64-70             */
	uint32_t cols[8];       /* their indices, LSB first                        */
	uint32_t early_exit;    /* 1: n<2 (:37-38); 2: pre-sorted (:60-62)        */
	uint32_t result_in_aux; /* 1 iff returned pointer == aux (:89,:92)        */
} rso_info;

size_t   rso_dtype_size(int dtype);

/* kdf(value) for one element, zero-extended to 64 bits (a8 in SURVEY.md 8a). */
uint64_t rso_kdf(const void *elem, int dtype, int order);

/* Upfront histogram exactly as loop 1 builds it: hist[256*j + digit_j] for
 * j < key_bytes (radix_sort.hpp:48-58).  hist must hold 256*8 entries. */
void rso_histogram(const void *src, size_t n, size_t rec_size, size_t key_off,
                   int dtype, int order, uint64_t *hist, uint64_t *n_unsorted);

/* radix_sort<T>(src, aux, n) with the default KDF (radix_sort.hpp:98-115).
 * Returns 0 when the result is in src, 1 when it is in aux. */
int rso_sort(void *src, void *aux, size_t n, int dtype, int order, rso_info *info);

/* radix_sort on records of rec_size bytes whose key (of kind dtype) sits at
 * byte key_off: the (T = struct, KeyFunc = field extract) instantiations of
 * radix_tests.cpp:41-43 and SURVEY.md's {f32 key, u32 payload} pin. */
int rso_sort_records(void *src, void *aux, size_t n, size_t rec_size, size_t key_off,
                     int dtype, int order, rso_info *info);

/* rs_sort_main with a caller-supplied Hist (radix_sort.hpp:28-33): the same sort, plus what the
 * reference leaves in the (pre-zeroed) histogram storage -- raw counts after the pre-sorted exit
 * (:48-62) and in skipped columns, end offsets (exclusive scan :72-80 + post-increments :85) in
 * kept columns, nothing for n < 2 -- as 256 * key_bytes uint64, reduced modulo 2^(8 hvt_bytes). */
int rso_sort_main_hist(void *src, void *aux, size_t n, int dtype, int order, int hvt_bytes,
                       uint64_t *hist_out, rso_info *info);

/* radix_sort_rank<T,IdxType> (radix_sort_rank.hpp:97-112) with the pass loop of
 * Listing 6 (radix_sort_u32_ranks.c:85-107: digit of src[idx[j]]), i.e. a
 * correct stable argsort.  index_buffer holds 2n entries of idx_bytes (1,2,4,8).
 * Returns 0 when the result is the first half, 1 when it is index_buffer + n. */
int rso_sort_rank(const void *src, size_t rec_size, size_t key_off, int dtype, int order,
                  void *index_buffer, int idx_bytes, size_t n, rso_info *info);

/* Same, but with the header's own pass loop (digit of src[j],
 * radix_sort_rank.hpp:80-89).  Only used to confirm where the reference's
 * defect does and does not show. */
int rso_sort_rank_asheader(const void *src, size_t rec_size, size_t key_off, int dtype, int order,
                           void *index_buffer, int idx_bytes, size_t n, rso_info *info);

/* Test helpers shared by the golden-vector generator and the tests. */
uint64_t rso_fnv1a64(const void *data, size_t bytes);
/* element i = low sizeof(T) bytes of (splitmix64() & mask), state starts at seed */
void rso_fill_splitmix(void *dst, size_t n, size_t elem_size, uint64_t seed, uint64_t mask);

#ifdef __cplusplus
}
#endif
#endif
