/*
 * rsx.h -- C ABI of the MI355X-native LSD radix sort (librsx.so).
 *
 * This is the drop-in boundary for the reference's hot path.  The reference has
 * no FFI: its surface is the header-only templates
 *
 *     T*       radix_sort(T* src, T* aux, size_t n, KeyFunc&& kf = basic_kdfs::kdf)
 *                                                           radix_sort.hpp:98-99
 *     IdxType* radix_sort_rank(const T* src, IdxType* index_buffer, size_t n, KeyFunc&& kf)
 *                                                           radix_sort_rank.hpp:97-98
 *     T*       rs_sort_main(src, aux, n, Hist&, kf)         radix_sort.hpp:31-32
 *
 * include/radix_sort.hpp, radix_sort_rank.hpp and radix_sort_basic_kdf.hpp in
 * this directory re-create those templates on top of the functions below
 * (see INTEGRATION.md).  All entry points use plain pointers and sizes only.
 *
 * Conventions
 *   - Return value: 0 on success, a negative RSX_E* code on failure;
 *     rsx_last_error() gives the message for the calling thread.  There is no
 *     CPU fallback: without a usable gfx950 device every sort call fails with
 *     RSX_ENODEVICE.
 *   - "Returned pointer" rule (radix_sort.hpp:89,:92): the result is in `src`
 *     when the number of kept (non-constant) 8-bit key columns is even, in
 *     `aux` when it is odd; n < 2 and already-sorted inputs return `src` and
 *     leave `aux` untouched (radix_sort.hpp:37-38,:60-62).  The buffer that is
 *     not returned has unspecified contents except in those early exits.
 *   - Order is a stable sort by kdf(key): output element images are bit-exact
 *     copies of the input's (NaN payloads, -0.0), radix_sort.hpp:85-87.
 *
 * Scratch memory.  The reference allocates nothing (radix_sort.hpp:98-115: two caller buffers, counters on the stack).
 *   This library keeps, per (device, stream) and until rsx_release / rsx_release_stream: status words of the look-back
 *   chains (1 KiB per 32 Ki-key tile and pass), and -- for the sorts without a histogram, which large evenly spread arrays
 *   take (rsx_info.hybrid == 5) -- slots: 256 x 1.25 n / 256 keys for the first MSB pass, of which the 204 that fit lie in the
 *   caller's SECOND buffer (keys-only sorts: the sample has proven the input unsorted before anything is written, so that
 *   buffer belongs to the sort as in the reference, radix_sort.hpp:82-92; the early exits leave it untouched as before) and
 *   the other 52 -- 0.25 n keys -- in scratch memory, and 65536 slots of 1.25 n / 65536 keys for the second pass (two bytes per
 *   key for 4-byte keys: 0.63 n keys' worth -- keys-only sorts of 4-byte keys up to 2^31 keys, round 5).  Key + payload and rank
 *   sorts (round 6): the level-1 slots of keys and payloads likewise in the spare buffers -- the second key and payload
 *   buffers; of a rank sort the index buffer's second half for the keys and its first half, until the ranks are written there,
 *   for the indices -- and, for the second pass, slots of two bytes per key and whole payloads in scratch memory (the two MSB
 *   digits are the slot, no leaf looks at more of a key than the two bytes below them).  A buffer's first allocation is
 *   what the sort needs; one that has to GROW grows in steps of an eighth of the next power of two (a slightly larger n does
 *   not reallocate again), the new one is allocated before the old one is freed, and nothing grows or is released inside a
 *   stream capture or a *_inplace_async call.  Measured (tools/footprint_probe.py, profiles/r06/footprint_probe.txt):
 *   2^28 u32 keys 1.10 GiB in all, 2^27 u64 keys 1.58, 2^28 u64 keys 3.14; 2^28 f32 keys + u32 payloads 2.42 GiB, -> u32
 *   ranks 2.42 (round 5: 5.5 for both, whole keys and all slots in scratch memory; RSX_NO_AUX_SLOTS=1: 4.41).  Round 6, 8-byte keys by
 *   (bit length, mantissa) digits (rsx_info.hybrid == 6): the level-1
 *   buckets lie in the caller's second buffer, scratch memory holds the level-2 slots -- four bytes per key, sized
 *   (1.125 n + 46 M) x 4 bytes for any distribution (2^28 keys: 1.4 GiB, of which BASELINE's Zipf-like keys use 0.8).
 *   If an allocation fails the sort takes the histogram-first
 *   route and the (device, stream) context does not ask again until rsx_reload_env() or rsx_release_stream();
 *   RSX_NO_BLIND=1 never asks.
 *
 * Environment switches (read ONCE, at the library's first call; rsx_reload_env() reads them again)
 *   RSX_VERIFY=1            after every scatter pass one tile is re-ranked without LDS
 *                           atomics and compared with the pass's output; a mismatch fails
 *                           the call (RSX_EVERIFY) -- at once for the blocking sorts (which
 *                           then keep to one pass per kept column), at rsx_verify_poll() or
 *                           the next blocking sort for the *_inplace_async ones.
 *   RSX_VERIFY=2            every blocking device sort runs as usual (any route, see rsx_info.hybrid)
 *                           and its RESULT is checked on the device (RSX_EVERIFY if not): keys only --
 *                           sorted, the input's key sum and key mix; key + payload -- no descent, the
 *                           input's key sum and a mix of its PAIRS; ranks -- a permutation of 0 .. n-1
 *                           through which the keys do not descend, equal keys in index order.
 *   RSX_FORCE_TABLE_RANK=1  use the table-ranked scatter kernel, which does not rely
 *                           on the lane order of returning LDS atomics (slower).
 *   RSX_NO_HYBRID=1         one scatter pass per kept column always (the reference's loop);
 *   RSX_NO_BLIND=1          every sort starts with the histogram (rsx_info.hybrid never 5);
 *   RSX_NO_LEAF_PREFIX=1    leaves of 8-byte keys sort by every column they have left;
 *   RSX_NO_DENSE_SLOTS=1    the second pass of such a sort writes whole keys into its slots;
 *   RSX_NO_LEAF16=1         its leaves are round 3's (two LDS passes per slot) instead of rsx_leaf16_kernel / rsx_leafk_kernel;
 *   RSX_NO_AUX_SLOTS=1      level-1 slots all in scratch memory; RSX_NO_NARROW_SLOTS=1: 8-byte keys always in whole-key slots;
 *   RSX_NO_NARROW_LEVEL1=1  8-byte keys: the level-1 slots always hold whole keys (round 6: low words where nothing below the
 *                           level-1 digit varies above bit 32 -- keys below 2^40 --, 12 + 8 + 12 instead of 16 + 12 + 12 bytes per key);
 *   RSX_LEAF16_MAXBIN=k     (tests) the fullest bin a leaf may have before it goes to those; RSX_NO_SHIFT=1: MSB digits on bytes only;
 *   RSX_NO_LEAF16Q=1        slots of up to 256 values take a wave per leaf instead of a row of sixteen lanes;
 *   RSX_NO_LEAF16W2K=1      slots of 1025 .. 2048 values take a 128-thread workgroup per leaf instead of a wave;
 *   RSX_NO_PASS16A=1        the level-2 pass of such a sort of 4-byte keys writes ragged runs (rsx_pass16_kernel) instead of whole
 *                           64-byte atoms (rsx_pass16a_kernel); RSX_NO_PASS32A=1: the level-1 pass is the chained kernel of round 4
 *                           (rsx_scatter2_kernel) instead of rsx_pass32a_kernel; RSX_NO_PASS16=1: so is the level-2 pass;
 *                           RSX_PASS16_WGS=1, RSX_PASS16_DBG=1|2: probes of rsx_pass16_kernel (DBG gives WRONG output);
 *   RSX_PASS32_PREFETCH=1   (probe) rsx_pass32a_kernel requests the next tile's keys while it writes the current one (default: when it
 *                           starts on the tile); RSX_PASS32_MIN_MI=k: that pass from k Mi keys on (default 52 / 24 Mi 4- / 8-byte keys);
 *   RSX_NO_LEAFC=1          no two-byte slots of more than 5120 values: keys-only sorts of 4-byte keys beyond 2^28 keys run as in
 *                           round 4 (whole-key slots up to 2^30 keys, one pass per column beyond; rsx_leafc.hpp);
 *   RSX_FORCE_LEAFC=1..6    (tests) two-byte slots of any size go through the leaves of the large ones: 1 the counting leaves,
 *                           2 .. 6 rsx_leaf16_kernel's 10240- / 20480- / 6144- / 7680- / 15360-value shape + the counting leaves' list launch;
 *   RSX_NO_PACKED_KEYS=1    rank sorts without a histogram go by byte columns only: neither the keys' packed varying bits nor, for
 *                           floats on a grid, their fixed-point integers are tried (SegCtl::compact);
 *   RSX_NO_UNSTABLE=1       the MSB passes of a sort without a histogram rank per wave (stable) as every other pass does;
 *   RSX_NO_LOG=1            8-byte keys never go by (bit length, mantissa) digits (rsx_info.hybrid never 6; rsx_logroute.hpp);
 *                           RSX_LOG_MIN_LOG2=k (tests): that route from 2^k keys on (default: from 24 Mi keys; at least 2^20);
 *                           RSX_LOG_LEAF_BIG=1 (tests): its leaves in the shape for 10240 values at every size;
 *   RSX_PAIRS_LEAF_BIG=1    (tests) key + payload and rank sorts without a histogram: the leaves' shape for slots of 10240 pairs (what
 *                           2^28 .. 2^29 pairs take, round 6) at every size;
 *   RSX_NO_PASS64A=1        the level-2 pass of 8-byte keys into four-byte slots is round 4's chained kernel (rsx_pass64.hpp);
 *   RSX_NO_ODD_STRIDE=1     the level-1 slots of a sort without a histogram lie 1.25 means apart as in round 5 (default: an odd
 *                           number of 64 KiB apart); RSX_CAP1_PAD_KIB=k (probe): k KiB more per level-1 slot;
 *   RSX_PROBE=bits          (measurements; results stay right) 1: the leaf table in reverse slot order, 4: every device-scheduled
 *                           sort as if called through rsx_sort_inplace_async_hint;
 *   RSX_NO_SLACK=1, RSX_TWO_LEVEL_MIN_LOG2=k, RSX_NO_SELF_PLAN=1, RSX_NO_FUSED_HIST=1
 *                           parts of the other routes.
 *   RSX_NO_SMALL_SORT, RSX_NO_HOST_SMALL, RSX_NO_FILL_RUNS, RSX_NO_SMALL_TILES, RSX_NO_SPECULATION,
 *   RSX_NO_NARROW_KEYS
 *                           switch single optimisations off (tests).
 *   RSX_COMPACT_BITS=1, RSX_HOST_REGISTER=1, RSX_ELEM_LOADS=1   opt-in variants.
 */
#ifndef RSX_H
#define RSX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Key kinds = the scalar types basic_kdfs::kdf accepts
 * (radix_sort_basic_kdf.hpp:19-46; radix_experiment.cpp:264-282). */
typedef enum rsx_dtype {
	RSX_U8 = 0, RSX_U16 = 1, RSX_U32 = 2, RSX_U64 = 3,
	RSX_I8 = 4, RSX_I16 = 5, RSX_I32 = 6, RSX_I64 = 7,
	RSX_F32 = 8, RSX_F64 = 9
} rsx_dtype;

/* RSX_ASCENDING = basic_kdfs::kdf; RSX_DESCENDING = its bitwise complement,
 * the reference's descending KDF (README.md:564-574, radix_tests.cpp:111-113,
 * :175-177).  Equal keys keep input order either way. */
typedef enum rsx_order { RSX_ASCENDING = 0, RSX_DESCENDING = 1 } rsx_order;

enum {
	RSX_OK = 0,
	RSX_EINVAL = -1,     /* bad dtype / size / null pointer                */
	RSX_ENODEVICE = -2,  /* no usable gfx950 device / HIP runtime failure  */
	RSX_ENOMEM = -3,     /* device workspace allocation failed             */
	RSX_EHIP = -4,       /* a HIP call failed (see rsx_last_error)         */
	RSX_EVERIFY = -5     /* RSX_VERIFY=1: a pass disagreed with its re-computation; RSX_VERIFY=2: a result is not the sorted input */
};

/* What the front half of rs_sort_main decided (radix_sort.hpp:48-80). */
typedef struct rsx_info {
	uint32_t key_bytes;     /* wc = sizeof(KeyType), radix_sort.hpp:40          */
	uint32_t ncols;         /* kept columns, radix_sort.hpp:64-70               */
	uint32_t cols[8];       /* their indices, LSB first                          */
	uint32_t early_exit;    /* 0 none; 1 n<2 (:37-38); 2 pre-sorted (:60-62)    */
	uint32_t result_in_aux; /* 1 iff the result is the second buffer / half     */
	uint32_t hybrid;        /* how the passes were made: 0 one per kept column (:82-90); 1 / 2: one / two passes by the
	                           highest kept column(s), then the remaining columns per bucket in LDS (README.md:647-650);
	                           3: one pass by the highest kept column, then one pass per remaining column inside its
	                           buckets; 4: as 2, the second pass written into per-bucket slots of a scratch array without
	                           counting first (evenly spread keys); 5: as 4 without the histogram -- a sample of the keys
	                           has PROVED the input unsorted and which columns are kept (the two facts radix_sort.hpp:60-70
	                           takes from the histogram; columns the sample found constant are checked on every key by
	                           the first pass), both passes write into slots, the bucket sizes come off the look-back
	                           chains (large arrays; blocking keys-only, rank and key + payload sorts);
	                           6: 8-byte keys whose magnitudes spread where their bytes do not (Zipf-like keys): the keys
	                           below 2^14 are counted and written out, the others go by (bit length, leading mantissa bits)
	                           into buckets of exactly counted sizes, by the next eight bits into slots, and through leaves
	                           (rsx_logroute.hpp; the pre-sorted exit and the kept columns come from that route's own
	                           one-read histogram kernel, exactly).  The
	                           result and the returned buffer are the same whichever it is. */
} rsx_info;

/* ---- environment ---------------------------------------------------------- */

int         rsx_device_count(void);   /* gfx950 devices visible; 0 if none / no runtime */
const char *rsx_last_error(void);     /* message of the calling thread's last failure   */
const char *rsx_version(void);
size_t      rsx_dtype_size(rsx_dtype dtype);
/* Bytes of device workspace a sort of n keys of this shape will hold on to. */
size_t      rsx_workspace_bytes(size_t n, rsx_dtype dtype, size_t payload_bytes);
/* Release every cached workspace / stream context of the calling process. */
void        rsx_release(void);
/* One context is kept per (device, stream) the library has been called with; a
 * program that creates many short-lived streams releases them one by one: frees the
 * workspace of (current device, stream) after synchronising the stream.  No other thread may be inside the library
 * on that (device, stream) while this runs, or start a call on it before it returns. */
void        rsx_release_stream(void *stream);

/* The RSX_* switches of the environment (diagnostics and A/B switches; every one is named where it acts, in
 * DESIGN.md) are read once, at the library's first call.  A process that changes them afterwards -- tests do -- calls
 * this to have them read again; RSX_FORCE_TABLE_RANK is only honoured before the first sort on a device.
 * NOT thread-safe against running sorts: the switches live in one process-wide record that every entry point reads
 * without a lock, so no other thread may be inside the library while this runs (as for rsx_release_stream). */
void        rsx_reload_env(void);

/* ---- radix_sort<T>(src, aux, n) -- radix_sort.hpp:98-115 ------------------ */

/* src/aux may be host pointers (staged over PCIe) or device pointers on the
 * current device (sorted in place in HBM); both must be of the same kind.
 * *result receives src or aux.  Blocking. */
int rsx_sort(void *src, void *aux, size_t n, rsx_dtype dtype, rsx_order order,
             void **result, rsx_info *info);

/* The same sort without any host synchronisation: every pass is scheduled on the
 * device (each finds its column, its buffers and "nothing to do" in the plan the
 * histogram kernels left in device memory), and the result always ends in d_buf --
 * when the number of kept columns is odd (radix_sort.hpp:89,:92 would return aux) a
 * last kernel copies it back.  d_scratch is the auxiliary buffer (n elements; its
 * contents afterwards are unspecified, untouched if the input was sorted).  Nothing
 * is returned but the status of the enqueue: the call can be captured into a HIP
 * graph (after one uncaptured call of the same size has sized the workspace) and
 * replayed on new contents of d_buf.
 * Round 4: the ROUTE is chosen on the device as well (rsx_async_route reports it): one MSB pass and leaves for mid-size
 * arrays; for large arrays (4-byte keys from 9 Mi keys, 8-byte keys from 8 Mi keys through this entry point; the blocking
 * rsx_sort_device starts at 7.5 Mi / 4.5 Mi) the sort without a histogram is
 * enqueued first and the histogram-first kernels behind it do nothing if it went through.  That attempt works in d_scratch
 * (once its sample has proven the input unsorted) and in slots in the (device, stream) workspace: 0.25 n + 0.625 n .. 1.25 n
 * keys of device memory (see "Scratch memory" above); rsx_sort_inplace_async_ws, whose state lies in the caller's
 * workspace, makes it when the workspace was sized by rsx_workspace_bytes_fast (below).  rsx_sort_pairs_inplace_async and rsx_sort_rank_inplace_async (4-byte keys with 4-byte
 * payloads / indices, 16 Mi .. 2^29 pairs) make the same attempt; rsx_async_route reports the last call's route for them too. */
int rsx_sort_inplace_async(void *d_buf, void *d_scratch, size_t n, rsx_dtype dtype, rsx_order order,
                           void *stream);
/* ... with what the caller knows about the keys.  RSX_HINT_EVEN_TOP_DIGITS: the caller has COUNTED the keys by their most
 * significant varying byte and found the counts even (no digit with twice its share) -- what the local sorts of a distributed
 * sort know from the split's gathered histogram.  Keys that an MSD split has put in order of that byte, piece by
 * piece, look clustered at each of the 64 places the sample of a sort without a histogram reads, and the sample would call the
 * attempt off: with the hint the level-1 digit's evenness is the caller's word, everything else is checked as always (the
 * level-2 digit in the sample; every slot's capacity by the passes themselves, so a wrong hint costs a lost attempt, never a
 * wrong result).  2^29 keys as rank 0 of 2 would receive them: 4.46 -> 3.44 ms (tools/presplit_probe.py). */
enum { RSX_HINT_EVEN_TOP_DIGITS = 1 };
int rsx_sort_inplace_async_hint(void *d_buf, void *d_scratch, size_t n, rsx_dtype dtype, rsx_order order,
                                void *stream, uint32_t hints);

/* LIFETIME of what a captured graph refers to.  The two *_inplace_async entry points
 * above and below keep their device state (flags, plan, histograms, status words of
 * the look-back) in the library's workspace of (current device, stream), which is
 * cached, GROWN -- freed and reallocated -- by a later larger sort on the same
 * (device, stream), and freed by rsx_release / rsx_release_stream.  A graph
 * captured from them is valid only until one of these happens, and its replays
 * must be ordered (stream order) against every other sort that uses the same
 * (device, stream) workspace.  A caller that keeps a graph should use the *_ws forms:
 * there all device state lies in d_workspace (256-byte aligned, at least
 * rsx_workspace_bytes(n, dtype, payload_bytes) bytes, owned by the caller for as
 * long as the graph lives; contents are scratch), so nothing the graph touches can
 * move, and graphs with different workspaces may replay on any streams. */
int rsx_sort_inplace_async_ws(void *d_buf, void *d_scratch, size_t n, rsx_dtype dtype, rsx_order order,
                              void *d_workspace, size_t workspace_bytes, void *stream);
/* Round 5: a workspace of rsx_workspace_bytes_fast(n, dtype) bytes (0.25 n + 0.63 n .. 1.25 n keys more than the minimum:
 * the slots of a sort without a histogram) lets rsx_sort_inplace_async_ws make that attempt INSIDE the workspace, so a graph
 * that is kept runs on the route of DESIGN.md 4c like every other sort (4- and 8-byte keys, 4 Mi .. 2^30 keys).  A smaller
 * workspace (at least rsx_workspace_bytes) works as before: histogram first.  rsx_async_route_ws reads the route the last sort
 * (or replay) in that workspace took (it waits for `stream`): 5, 1 or 0 as rsx_async_route. */
size_t rsx_workspace_bytes_fast(size_t n, rsx_dtype dtype);
int rsx_async_route_ws(const void *d_workspace, size_t workspace_bytes, size_t n, rsx_dtype dtype, void *stream, uint32_t *route);
int rsx_sort_pairs_inplace_async_ws(void *d_keys, void *d_keys_scratch, void *d_vals, void *d_vals_scratch,
                                    size_t n, rsx_dtype dtype, size_t payload_bytes, rsx_order order,
                                    void *d_workspace, size_t workspace_bytes, void *stream);

/* Key + payload in the same manner: both arrays sorted in place by the keys (stable),
 * never a host synchronisation, HIP-graph capturable.  payload_bytes: 4 or 8. */
int rsx_sort_pairs_inplace_async(void *d_keys, void *d_keys_scratch, void *d_vals, void *d_vals_scratch,
                                 size_t n, rsx_dtype dtype, size_t payload_bytes, rsx_order order,
                                 void *stream);

/* radix_sort_rank (radix_sort_rank.hpp:97-112) in the same manner: the stable argsort of d_src (untouched) without a
 * host synchronisation, HIP-graph capturable.  The ranks ALWAYS end in the first half of d_index_buffer (2n entries of
 * idx_bytes = 4 or 8 bytes): the number of passes is only known on the device, where the passes take their buffers from
 * the plan so that the last one writes the first half; sorted keys give 0 .. n-1 there (:52,:55-57).  The keys' work
 * copies live in the (device, stream) workspace: call it once outside a capture so that the workspace has its size.  Keys
 * travel at full width here (the blocking rsx_sort_rank_device hands on only the bytes still to be sorted by). */
int rsx_sort_rank_inplace_async(const void *d_src, void *d_index_buffer, size_t n, rsx_dtype dtype,
                                size_t idx_bytes, rsx_order order, void *stream);

/* RSX_VERIFY=1 also checks one tile of every pass of the *_inplace_async sorts (re-ranked with ballots on the device, as
 * for the blocking sorts), but those never wait: their mismatches add up on the device.  This waits for `stream`, reports
 * the count (RSX_EVERIFY if it is not zero) and resets it; the next blocking sort on the stream does the same. */
int rsx_verify_poll(void *stream, uint64_t *mismatches);

/* The route the LAST rsx_sort_inplace_async on (current device, stream) took -- the device decides it and the call never
 * waits, so it can only be asked for afterwards: this waits for `stream` and reads the device's words back.  *route as
 * rsx_info.hybrid: 5 = no histogram, two MSB passes into slots and leaves; 1 = one MSB pass and leaves; 0 = histogram and one
 * pass per kept column (also: sorted input, and the one-launch sort of small arrays).  The loop all of them stand for is
 * radix_sort.hpp:82-90.  Replaying a captured graph of such a sort re-decides the route on the device each time; what this
 * reports is the last replay's.  The words it reads are the context's: a BLOCKING sort on the same stream between the
 * device-scheduled sort and this call overwrites them (ask before that sort, or use another stream for it). */
int rsx_async_route(void *stream, uint32_t *route);

/* rs_sort_main / rs_sort_rank with a caller-supplied Hist (radix_sort.hpp:28-33,
 * radix_sort_rank.hpp:22-23): arms the CALLING THREAD's next blocking sort call
 * (rsx_sort, rsx_sort_device, rsx_sort_rank*, rsx_sort_pairs_device,
 * rsx_sort_records*) to also write the counts of loop 1 (radix_sort.hpp:48-58):
 * hist[256 * j + d] = number of keys whose KDF byte j is d, j < key bytes;
 * `entries` = room in hist (>= 256 * key bytes, or the sort fails with
 * RSX_EINVAL).  One shot: the sort disarms it; hist == NULL disarms by hand.
 * Sorts of n < 2 keys and the *_async entry points never write.  From the counts
 * and rsx_info the template wrapper reproduces what the reference leaves in the
 * caller's storage (include/radix_sort.hpp, hist_post_state). */
int rsx_capture_histogram(uint64_t *hist, size_t entries);

/* Device-resident variant for callers that own a HIP stream (`stream` is a
 * hipStream_t, NULL = the default stream).  The column plan has to reach the
 * host to apply the returned-pointer rule, so the call synchronises `stream`
 * once after the histogram pass; the scatter passes are enqueued and NOT
 * waited for: the result is valid for work ordered after them on `stream`. */
int rsx_sort_device(void *d_src, void *d_aux, size_t n, rsx_dtype dtype, rsx_order order,
                    void *stream, void **result, rsx_info *info);

/* ---- key + payload (struct-of-arrays) ------------------------------------- */

/* Stable sort of (key[i], payload[i]) pairs by kdf(key): the device form of the
 * reference's radix_sort on {key, payload} records (radix_tests.cpp:41-56 shape,
 * SURVEY.md 8d cfg 4).  payload_bytes is 4 or 8.  Keys and payloads ping-pong
 * together; info->result_in_aux tells which pair of buffers holds the result. */
int rsx_sort_pairs_device(void *d_keys, void *d_keys_aux, void *d_vals, void *d_vals_aux,
                          size_t n, rsx_dtype dtype, size_t payload_bytes, rsx_order order,
                          void *stream, rsx_info *info);

/* ---- radix_sort_rank<T,IdxType>(src, index_buffer, n) --------------------- */

/* radix_sort_rank.hpp:97-112 with the pass semantics of Listing 6
 * (radix_sort_u32_ranks.c:85-107), i.e. a correct stable argsort: see
 * SURVEY.md section 4 for the header's defect.  index_buffer holds 2n indices of
 * idx_bytes (1, 2, 4 or 8; n must fit).  *result receives index_buffer (kept
 * columns even, n<2, pre-sorted) or index_buffer + n (odd).  Pre-sorted input
 * writes 0..n-1 to the first half and leaves the second untouched
 * (radix_sort_rank.hpp:52,:55-57); n == 1 writes index_buffer[0] = 0 (:28-32).
 * Host or device pointers (both of the same kind).  Blocking. */
int rsx_sort_rank(const void *src, void *index_buffer, size_t n, rsx_dtype dtype,
                  size_t idx_bytes, rsx_order order, void **result, rsx_info *info);

/* Device-resident variant; idx_bytes is 4 or 8. */
int rsx_sort_rank_device(const void *d_src, void *d_index_buffer, size_t n, rsx_dtype dtype,
                         size_t idx_bytes, rsx_order order, void *stream, void **result,
                         rsx_info *info);

/* ---- records with a host-evaluated key (opaque KeyFunc) -------------------- */

/* radix_sort(src, aux, n, kf) for trivially copyable T and an arbitrary host
 * KeyFunc (radix_tests.cpp:41-43,:111-113; README.md:562-591): the template
 * wrapper evaluates kf once per element into `keys` (n unsigned keys of
 * key_bytes = sizeof(KeyType) in {1,2,4,8}, already KDF-applied); the device
 * rank-sorts the keys and gathers the rec_bytes-sized records.  Host pointers.
 * *result receives src or aux by the returned-pointer rule. */
int rsx_sort_records(void *src, void *aux, size_t n, size_t rec_bytes,
                     const void *keys, size_t key_bytes, void **result, rsx_info *info);

/* ---- records with a declared key (tagged KeyFunc) ----------------------------- */

/* radix_sort(src, aux, n, kf) where kf is "the scalar field at byte key_offset
 * of the element, default KDF of its type, ascending or complemented": the shapes
 * of radix_tests.cpp:41-43,:45-69,:111-113 and README.md:562-591 written as data
 * instead of code (include/radix_sort.hpp: rsx_kdf::by_member<&T::field>).
 * Key extraction, stable rank sort and the gather of the rec_bytes-sized records
 * all run on the device.  *result receives src or aux by the returned-pointer rule
 * (radix_sort.hpp:89,:92); pre-sorted input returns src with aux untouched. */
int rsx_sort_records_tagged(void *src, void *aux, size_t n, size_t rec_bytes, size_t key_offset,
                            rsx_dtype key_dtype, rsx_order order, void **result, rsx_info *info);

/* Device-pointer form: records resident in HBM, work enqueued on `stream`
 * (one synchronisation inside, as rsx_sort_device). */
int rsx_sort_records_tagged_device(void *d_src, void *d_aux, size_t n, size_t rec_bytes,
                                   size_t key_offset, rsx_dtype key_dtype, rsx_order order,
                                   void *stream, void **result, rsx_info *info);

/* Same derivation for radix_sort_rank with an opaque KeyFunc. */
int rsx_sort_rank_keys(const void *keys, size_t key_bytes, void *index_buffer, size_t n,
                       size_t idx_bytes, void **result, rsx_info *info);

/* ---- building blocks exported for the multi-GPU path and the tests --------- */

/* Upfront histogram of every 8-bit KDF column (radix_sort.hpp:48-58):
 * d_hist receives 256 * key_bytes uint64 counts (device memory), *d_unsorted a
 * non-zero uint32 iff some kdf(key[i]) > kdf(key[i+1]).  Enqueued on stream. */
int rsx_histogram_device(const void *d_src, size_t n, rsx_dtype dtype, rsx_order order,
                         uint64_t *d_hist, uint32_t *d_unsorted, void *stream);

/* The MSD split as one ordinary stable scatter pass by the top KDF byte itself
 * (256 digits; README.md:647-650): d_dst = d_src ordered by that
 * byte, top_hist (host, 256 uint64) = its counts.  Destinations that are
 * contiguous ranges of the byte are contiguous ranges of d_dst:
 * digit d occupies [sum(top_hist[0..d)), +top_hist[d]).  The counts are ready on
 * return; the pass is enqueued on stream.  column < 0: the top byte; otherwise
 * the byte to split by (a caller that knows the bytes above it to be constant). */
int rsx_msd_split_device(const void *d_src, void *d_dst, size_t n, rsx_dtype dtype,
                         rsx_order order, int column, uint64_t *top_hist, void *stream);

/* The same pass for a caller that already has the shard's column counts -- d_hist: the 256 * key_bytes uint64 counts
 * rsx_histogram_device left in device memory (one read of the shard gives every column's, so the column to split by can be
 * chosen from them without a trial pass).  Counts nothing, waits for nothing: everything is only enqueued on `stream`.
 * column | RSX_SPLIT_HOT: the caller, who has the counts, says that one digit of the column holds an eighth of the shard's
 * keys or more -- the pass then ranks its dominant digits by ballots (what rsx_msd_split_device decides from its own counts). */
#define RSX_SPLIT_HOT 0x100
int rsx_msd_split_async(const void *d_src, void *d_dst, size_t n, rsx_dtype dtype, rsx_order order,
                        int column, const uint64_t *d_hist, void *stream);

/* radix_sort(src, aux, n, kdf) on HOST buffers with the work spread over `ndev`
 * devices of this one process (radix_sort.hpp:98-115 semantics: early exits,
 * kept columns, returned pointer, `aux` untouched on sorted input -- decided
 * globally, exactly as one rsx_sort call would).  devices: HIP device indices, one
 * per rank; an index may repeat (several ranks then share a device).  Shard r =
 * the r-th n/ndev-th of src; MSD split by the highest kept byte, peer-to-peer
 * exchange of the digit ranges, local LSD sorts, each written to its place in
 * the result buffer.  Blocking.  The one-process-per-GPU form of the same
 * algorithm (RCCL all-to-all-v) is radix_sorting_amd/multi.py. */
int rsx_sort_multi(void *src, void *aux, size_t n, rsx_dtype dtype, rsx_order order,
                   const int *devices, int ndev, void **result, rsx_info *info);

/* ---- measurement hooks and input generator (bench.py, tests) ------------------ */

/* Between rsx_profile_begin() and rsx_profile_end() every histogram and scatter
 * kernel the library launches is bracketed by HIP events recorded on the stream
 * the kernel is launched on; rsx_profile_end() waits for them and returns the
 * summed kernel durations.  scatter_bytes = sum over scatter launches of
 * n * 2 * (key bytes + payload bytes), the algorithmic traffic of one pass
 * (SURVEY.md 8d); hist_bytes = sum of n * key bytes. */
typedef struct rsx_profile {
	double   hist_ms;
	double   scatter_ms;
	uint64_t hist_launches;
	uint64_t scatter_launches;
	uint64_t hist_bytes;
	uint64_t scatter_bytes;
	double   leaf_ms;        /* rsx_leaf_sort_kernel (sorts that take one MSB pass and leaves): n * 2 * key bytes per launch
	                            (n * (2 + key bytes) where the leaves read two-byte slots) */
	uint64_t leaf_launches;
	uint64_t leaf_bytes;
	double   narrow_ms;      /* scatter passes that write their keys narrowed into slots (a different instantiation of the
	                            pass kernel, counted apart from scatter_*): n * (key bytes + 2) per launch */
	uint64_t narrow_launches;
	uint64_t narrow_bytes;
	double   called_off_ms;  /* launches that did nothing, or whose output was discarded: the kernels of a sort without a
	                            histogram that its sample or an overflowing slot called off, leaves enqueued for a route the
	                            plan did not choose.  Their time is here, their bytes are in none of the byte counts; the
	                            narrowed level-2 pass and leaves of 8-byte keys are booked with the four-byte slots the device
	                            chose (blocking sorts: the host knows the verdict; the *_inplace_async sorts never learn it
	                            and book what they enqueued). */
	uint64_t called_off_launches;
} rsx_profile;
int rsx_profile_begin(void);
int rsx_profile_end(rsx_profile *out);

/* d_dst[i] = low elem_bytes bytes of (splitmix64 output number first_index + i
 * of the stream seeded with `seed`) & mask: the generator of SURVEY.md section 4
 * and 8d, evaluated counter-based on the device (element i of the serial
 * generator is finalizer(seed + (i+1) * 0x9E3779B97F4A7C15)). */
int rsx_fill_splitmix_device(void *d_dst, size_t n, size_t elem_bytes, uint64_t seed,
                             uint64_t mask, uint64_t first_index, void *stream);

/* Keeps `stream` busy for about `microseconds` (one spinning wave): a known-length
 * occupant for stream / hardware-queue experiments (multi.py picks the stream whose
 * kernels overlap RCCL's with it). */
int rsx_spin_device(uint64_t microseconds, void *stream);

#ifdef __cplusplus
}
#endif
#endif
