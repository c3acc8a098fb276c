// rsx_kernels.hpp -- CDNA4 (gfx950) kernels of the LSD radix sort: everything but the default scatter pass.
//
// The three loops of the reference's rs_sort_main (radix_sort.hpp:31-93) become
//
//   rsx_hist_kernel       loop 1 (:48-58): every 8-bit column's histogram in ONE read of the keys + the pre-sorted
//   + rsx_hist_reduce_    test.  Histograms are privatised in LDS per workgroup (R lane-striped copies per bin so
//     kernel              that equal digits do not serialise on one LDS address); each workgroup writes its counts to
//                         its own row and a split reduce adds the rows up (up to 128 workgroups add them to the
//                         histogram themselves instead).
//   rsx_plan_kernel       column-skip probe (:64-70) + exclusive scan (:72-80): the 256-bin scan of a column is done
//                         by one 64-lane wavefront through LDS (4 bins per lane); the column's frequent digits for
//                         the HOT scatter kernels; the block that finishes last writes the plan.
//   rsx_scatter2_kernel   one scatter pass (:82-90), the default: rsx_scatter2.hpp.
//   rsx_scatter_kernel    the same pass without relying on the lane order of returning LDS atomics (the fallback when
//                         the device self-check fails): a workgroup takes a tile (ticket order), ranks its keys inside
//                         each wavefront through per-wave LDS match tables + mbcnt/popcount, chains the per-digit tile
//                         offsets with a decoupled look-back over agent-scope status words, stages the tile in LDS in
//                         output order and writes coalesced runs.
//   rsx_small_sort_kernel, rsx_small_pairs_kernel   the whole sort in one workgroup for small arrays: rsx_small.hpp.
//   rsx_fill_runs_kernel  keys only, ONE kept column (1-byte keys; wider keys that differ in one byte): no scatter at all --
//                         the sorted array is that column's histogram written out.
//   rsx_joint16_kernel, rsx_joint16_scan_kernel, rsx_fill16_kernel   keys only, 2-byte keys with both columns kept: one
//                         16-bit digit -- the joint histogram of the two bytes, scanned and written out.
//
// Also here: the key derivation (kdf_apply), the status-word format of the look-back chain, and the small helper
// kernels (fill, iota, convert, key extraction, record gather).
//
// Stability: a wave owns a contiguous slice of the tile, lane l of round r holds element slice + 64 r + l, rounds
// are ranked in order and lanes in lane order, waves and tiles are prefix-summed in memory order -- so equal digits
// keep their input order exactly as the reference's in-order traversal with post-increment does.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rsx {

typedef unsigned long long u64;
typedef unsigned int u32;
typedef u32 u32x4 __attribute__((ext_vector_type(4)));
typedef u32 u32x2 __attribute__((ext_vector_type(2)));

struct NoVal {};  // "keys only" payload tag

template <typename T> struct val_bytes { static constexpr int value = sizeof(T); };
template <> struct val_bytes<NoVal> { static constexpr int value = 0; };

#define RSX_COMPILER_FENCE() asm volatile("" ::: "memory")

// ---- key derivation ---------------------------------------------------------
// radix_sort_basic_kdf.hpp:19-46 folded into three per-launch constants so one
// kernel serves unsigned / signed / float keys and both orders:
//   unsigned  fmask = 0   sflip = 0        kdf = k
//   signed    fmask = 0   sflip = highbit  kdf = k ^ highbit            (:26-30)
//   float     fmask = ~0  sflip = highbit  kdf = k ^ (-(k>>31) | 1<<31) (:32-46)
//   desc = ~0 complements the result (README.md:564-574).
template <typename KT> struct KdfArgs { KT fmask, sflip, desc; };

template <typename KT>
__device__ __forceinline__ KT kdf_apply(KT raw, const KdfArgs<KT> a)
{
	typedef typename std::make_signed<KT>::type ST;
	const KT sign = (KT)((ST)raw >> (sizeof(KT) * 8 - 1));  // all ones iff top bit set
	return (KT)(raw ^ ((sign & a.fmask) | a.sflip) ^ a.desc);
}

// popcount of the bits of `m` below this lane
__device__ __forceinline__ u32 mbcnt64(u64 m)
{
	return __builtin_amdgcn_mbcnt_hi((u32)(m >> 32), __builtin_amdgcn_mbcnt_lo((u32)m, 0u));
}

// =============================================================================
// Kernel 1: histogram of all columns + pre-sorted test: rsx_hist.hpp
// =============================================================================

// =============================================================================
// Kernel 2: column-skip probe + exclusive scan
// =============================================================================

struct Plan {            // 64 bytes, copied to the host after the plan kernels
	u32 ncols;           // kept columns (radix_sort.hpp:64-70)
	u32 sorted;          // 1: pre-sorted early exit (radix_sort.hpp:60-62)
	u32 cols[8];
	u32 hot;             // bit c: one digit of column c holds an eighth of the keys or more
	u32 vary_lo, vary_hi;   // the bits of the KDF key that are not the same in all keys (byte c from column c's histogram); 0: not computed
	u32 hyb;             // rsx_hybrid.hpp: HYB_NONE (one pass per kept column), HYB_ONE_LEVEL, HYB_TWO_LEVEL
	u32 max1;            // the largest bin of the highest kept column (the largest level-1 bucket)
	u32 pad;
};

// What rsx_plan_kernel may choose (rsx_hybrid.hpp); all zero: one pass per kept column, as the reference.
struct HybCaps {
	u32 cap1;            // one level: every bucket of the highest kept column holds at most this many keys (0: never)
	u32 cap2;            // two levels: the estimate for the largest (digit, digit) bucket is at most this (0: never)
	u32 min_cols1, min_cols2;   // kept columns needed for either
};

}  // namespace rsx
#include "rsx_hist.hpp"
namespace rsx {

// Exclusive scan of 256 u64 values held in LDS, by ONE wavefront (lanes 0..63 of the caller):
// 4 bins per lane, lane totals combined with a Hillis-Steele pass through LDS.
__device__ __forceinline__ void wave_scan_256(u64 *vals, u64 *lsum, u32 lane)
{
	u64 c[4];
#pragma unroll
	for (int i = 0; i < 4; ++i)
		c[i] = vals[4 * lane + i];
	const u64 mine = c[0] + c[1] + c[2] + c[3];
	u64 incl = mine;
	lsum[lane] = incl;
#pragma unroll
	for (int off = 1; off < 64; off <<= 1) {   // DS operations of one wave execute in issue order
		RSX_COMPILER_FENCE();
		const u64 add = lane >= (u32)off ? lsum[lane - off] : 0;
		RSX_COMPILER_FENCE();
		incl += add;
		lsum[lane] = incl;
	}
	u64 a = incl - mine;
#pragma unroll
	for (int i = 0; i < 4; ++i) {
		vals[4 * lane + i] = a;
		a += c[i];
	}
}

// One workgroup (256 threads, thread = digit) per column.  ghist[col][256] holds counts on entry; on exit the
// exclusive offset of the digit: every key with a smaller digit -- the reference's exclusive scan
// (radix_sort.hpp:72-80).  kept[col] answers the column-skip probe (:64-70).
__device__ __forceinline__ void plan_finish(const u32 *kept, u32 wc, const u32 *unsorted, Plan *plan, Plan *host_plan, u64 n,
                                            HybCaps caps);

template <typename KT>
__global__ __launch_bounds__(256) void rsx_plan_kernel(const KT *__restrict__ src, u64 n, u64 *__restrict__ ghist,
                                                       KdfArgs<KT> ka, u32 *__restrict__ kept,
                                                       u32 *__restrict__ hotd = nullptr, u32 *done = nullptr,
                                                       const u32 *unsorted = nullptr, Plan *plan = nullptr,
                                                       Plan *host_plan = nullptr, HybCaps caps = HybCaps{0, 0, 0, 0})
{
	constexpr int WC = sizeof(KT);
	__shared__ u64 tot[256];
	__shared__ u64 lsum[64];
	const u32 d = threadIdx.x, col = blockIdx.x;
	u64 *h = ghist + 256 * col + d;

	const u64 total = *h;
	const KT key0 = kdf_apply(src[0], ka);                         // radix_sort.hpp:65
	if (d == ((u32)(key0 >> (8 * col)) & 0xFFu))
		kept[col] = total != n;                                    // radix_sort.hpp:67
	if (total >= n / 8 + 1)
		atomicOr(&kept[8 + col], 1u);                              // a hot digit (see Plan::hot)
	// the bits of this column that vary: a bit varies iff the digits that occur do not agree in it (README.md:716-758's
	// bit mask, read off the histogram instead of the keys)
	__shared__ u32 s_and, s_or, s_max;
	if (d == 0) {
		s_and = 0xFFu;
		s_or = 0;
		s_max = 0;
	}
	tot[d] = total;
	__syncthreads();
	if (total) {
		atomicAnd(&s_and, d);
		atomicOr(&s_or, d);
		atomicMax(&s_max, total > 0xFFFFFFFFull ? 0xFFFFFFFFu : (u32)total);
	}
	__syncthreads();
	if (d == 0) {
		atomicOr(&kept[8 + col], ((s_and ^ s_or) & 0xFFu) << 8);
		kept[16 + col] = s_max;                                    // the column's largest bin (rsx_hybrid.hpp)
	}
	__syncthreads();
	// hotd[col]: up to four digits that hold a sixteenth of the keys or more, most frequent first, a byte each; hotd[8]:
	// bit 4 col + r = slot r of column col is valid (all zeroed by the caller).  The HOT scatter kernels rank these
	// digits with ballots instead of LDS atomics (rsx_scatter2.hpp).
	if (hotd && total >= n / 16 + 1) {
		u32 rank = 0;
		for (u32 e = 0; e < 256; ++e)
			rank += (tot[e] > total || (tot[e] == total && e < d)) ? 1u : 0u;
		if (rank < 4) {
			atomicOr(&hotd[col], d << (8 * rank));
			atomicOr(&hotd[8], 1u << (4 * col + rank));
		}
	}
	__syncthreads();
	if (d < 64)
		wave_scan_256(tot, lsum, d);                               // radix_sort.hpp:74-79
	__syncthreads();
	*h = tot[d];
	// `done` (zeroed by the caller): the block that finishes last writes the plan (no launch of its own for that)
	if (done) {
		__shared__ u32 s_last;
		__syncthreads();
		if (d == 0) {
			__threadfence();
			s_last = atomicAdd(done, 1u) == gridDim.x - 1 ? 1u : 0u;
		}
		__syncthreads();
		if (s_last && d == 0) {
			__threadfence();
			plan_finish((const u32 *)kept, WC, unsorted, plan, host_plan, n, caps);
		}
	}
}

// `host_plan`: the same 64 bytes in pinned, device-visible host memory -- the host reads the plan there once this kernel
// has completed, without a copy of its own (one launch less per sort).
__device__ __forceinline__ void plan_finish(const u32 *kept, u32 wc, const u32 *unsorted, Plan *plan, Plan *host_plan, u64 n,
                                            HybCaps caps)
{
	// (no local Plan: an array indexed by a run-time count would live in scratch memory, and a kernel with scratch is
	// dispatched noticeably more slowly -- this runs at the end of the FUSED histogram kernel of every small sort)
	u64 colpack = 0;   // 4 bits per kept column, LSB first (radix_sort.hpp:66-69)
	u32 nc = 0;
	for (u32 i = 0; i < wc; ++i)
		if (kept[i]) {
			colpack |= (u64)i << (4 * nc);
			++nc;
		}
	const u32 sorted = *unsorted == 0;                             // radix_sort.hpp:60
	u32 hot = 0, vary_lo = 0, vary_hi = 0;
	for (u32 i = 0; i < wc; ++i) {
		hot |= (kept[8 + i] & 1u) << i;
		const u32 v = (kept[8 + i] >> 8) & 0xFFu;
		if (i < 4)
			vary_lo |= v << (8 * i);
		else
			vary_hi |= v << (8 * (i - 4));
	}
	// One MSB pass and leaves (rsx_hybrid.hpp) where the keys spread over the digits of their top kept column(s): decided
	// here, from the histograms, so that the device-scheduled first pass already takes the right column.
	u32 hyb = 0;
	const u32 top = nc ? (u32)(colpack >> (4 * (nc - 1))) & 15u : 0u;
	const u32 max1 = nc ? kept[16 + top] : 0;
	// A column with a dominant digit (Plan::hot: one bin holds an eighth of the keys or more) among those the LEAVES would sort
	// by keeps the sort on the pass kernels: 64 lanes on one LDS counter make a leaf column ten times slower, and the pass
	// kernels have their ballot-ranked HOT form for such columns.  (A dominant digit in a column the MSB passes go by shows
	// in the bucket sizes and is excluded by them.)
	u32 keptmask = 0;
	for (u32 i = 0; i < nc; ++i)
		keptmask |= 1u << ((u32)(colpack >> (4 * i)) & 15u);   // (a skipped column is one bin with all n keys: not meant here)
	const u32 hot_below1 = nc ? hot & keptmask & ~(1u << top) : 0u;
	const u32 second = nc >= 2 ? (u32)(colpack >> (4 * (nc - 2))) & 15u : 0u;
	const u32 hot_below2 = hot_below1 & ~(1u << second);
	if (!sorted && n < (1ull << 30)) {
		if (caps.cap1 && nc >= caps.min_cols1 && max1 <= caps.cap1 && !hot_below1) {
			hyb = 1;
		} else if (caps.cap2 && nc >= caps.min_cols2 && nc >= 2 && !hot_below2) {
			// the largest (digit, digit) bucket if the two top columns were independent; rsx_seg_plan_kernel has the last word
			const u64 est = (u64)max1 * kept[16 + second] / n;
			if (est <= caps.cap2 - caps.cap2 / 4)
				hyb = 2;
		}
	}
	plan->hyb = host_plan->hyb = hyb;
	plan->max1 = host_plan->max1 = max1;
	plan->ncols = host_plan->ncols = nc;
	plan->sorted = host_plan->sorted = sorted;
	plan->hot = host_plan->hot = hot;
	plan->vary_lo = host_plan->vary_lo = vary_lo;
	plan->vary_hi = host_plan->vary_hi = vary_hi;
	for (u32 i = 0; i < 8; ++i) {
		const u32 c = i < nc ? (u32)(colpack >> (4 * i)) & 15u : 0u;
		plan->cols[i] = c;
		host_plan->cols[i] = c;
	}
	__threadfence_system();
}


// rsx_plan_kernel's work for every column in ONE workgroup of 1024 threads, four columns at a time, thread (g, d) = digit d
// of column col0 + g; the plan is finished from LDS.  (rsx_plan_kernel, one workgroup per column, ends with "the workgroup
// that finishes last writes the plan": a device-scope fence per workgroup and a chain of dependent global loads in one
// thread -- 10 us for four columns; this kernel takes about 4.)
template <typename KT>
__global__ __launch_bounds__(1024) void rsx_plan_all_kernel(const KT *__restrict__ src, u64 n, u64 *__restrict__ ghist, KdfArgs<KT> ka,
                                                            u32 *__restrict__ kept_out, u32 *__restrict__ hotd,
                                                            const u32 *unsorted, Plan *plan, Plan *host_plan, HybCaps caps,
                                                            const u32 *gate = nullptr)
{
	if (gate && *gate == GATE_DONE)   // (a device-scheduled sort that got by without the histogram: its plan stands)
		return;
	constexpr int WC = sizeof(KT);
	__shared__ u64 tot[4][256];
	__shared__ u64 lsum[4][64];
	__shared__ u32 s_and[4], s_or[4], s_max[4];
	__shared__ u32 s_kept[24];     // [col] kept, [8 + col] hot bit | varying bits << 8, [16 + col] largest bin
	const u32 g = threadIdx.x >> 8, d = threadIdx.x & 255u;
	if (threadIdx.x < 24)
		s_kept[threadIdx.x] = 0;
	const KT key0 = kdf_apply(src[0], ka);                         // radix_sort.hpp:65
	const u32 uns = *unsorted;
	__syncthreads();
	for (u32 col0 = 0; col0 < (u32)WC; col0 += 4) {
		const u32 col = col0 + g;
		const bool on = col < (u32)WC;
		u64 *h = ghist + 256 * (on ? col : 0) + d;
		const u64 total = on ? *h : 0;
		if (on && d == ((u32)(key0 >> (8 * col)) & 0xFFu))
			s_kept[col] = total != n;                              // radix_sort.hpp:67
		if (on && total >= n / 8 + 1)
			atomicOr(&s_kept[8 + col], 1u);                        // a hot digit (see Plan::hot)
		if (d == 0) {
			s_and[g] = 0xFFu;
			s_or[g] = 0;
			s_max[g] = 0;
		}
		tot[g][d] = total;
		__syncthreads();
		if (on && total) {
			atomicAnd(&s_and[g], d);
			atomicOr(&s_or[g], d);
			atomicMax(&s_max[g], total > 0xFFFFFFFFull ? 0xFFFFFFFFu : (u32)total);
		}
		__syncthreads();
		if (on && d == 0) {
			atomicOr(&s_kept[8 + col], ((s_and[g] ^ s_or[g]) & 0xFFu) << 8);
			s_kept[16 + col] = s_max[g];
		}
		if (on && hotd && total >= n / 16 + 1) {
			u32 rank = 0;
			for (u32 e = 0; e < 256; ++e)
				rank += (tot[g][e] > total || (tot[g][e] == total && e < d)) ? 1u : 0u;
			if (rank < 4) {
				atomicOr(&hotd[col], d << (8 * rank));
				atomicOr(&hotd[8], 1u << (4 * col + rank));
			}
		}
		__syncthreads();
		if (d < 64)
			wave_scan_256(tot[g], lsum[g], d);                         // radix_sort.hpp:74-79
		__syncthreads();
		if (on)
			*h = tot[g][d];
		__syncthreads();
	}
	if (threadIdx.x < 24)
		kept_out[threadIdx.x] = s_kept[threadIdx.x];
	if (threadIdx.x == 0)
		plan_finish((const u32 *)s_kept, WC, &uns, plan, host_plan, n, caps);
}

// =============================================================================
// Kernel 3: one stable scatter pass (onesweep)
// =============================================================================

template <typename ST> struct StatusBits;
template <> struct StatusBits<u32> {
	static constexpr u32 SHIFT = 30;
	static constexpr u32 VALMASK = (1u << 30) - 1;
};
template <> struct StatusBits<u64> {
	static constexpr u32 SHIFT = 62;
	static constexpr u64 VALMASK = (1ull << 62) - 1;
};
enum : u32 { ST_EMPTY = 0, ST_AGGREGATE = 1, ST_PREFIX = 2 };

enum : u32 {
	SCATTER_GEN_INDEX = 1,   // payload of element i is i (first rank pass, radix_sort_rank.hpp:52)
	SCATTER_SKIP_KEYS = 2,   // do not write keys (last rank pass: only the indices are wanted)
	SCATTER_COL_SHIFT = 12,  // bits 12-14: the pass's column (HOT kernels: which word of hotd; the shift no longer tells once a rank sort has narrowed its keys)
	SCATTER_ONE_COL_FILLED = 8, // device-scheduled pass 0 of a keys-only sort: do nothing if ONE column is kept (rsx_fill_runs_kernel writes the result)
	SCATTER_HOT = 16,        // host side only: one digit holds an eighth of the keys or more -> the HOT kernels (rsx_scatter2.hpp)
	SCATTER_ELEM_LOADS = 32, // whole tiles are read with element loads instead of 16-byte loads + a transposition through the LDS
	SCATTER_DBG_LINEAR = 64, // probe only: write the staged tile back to its own position (no scatter)
	SCATTER_DBG_NOSTORE = 128, // probe only: skip the global stores
	SCATTER_DBG_NOLOADB = 256, // probe only: phase B fabricates keys instead of re-reading them
	SCATTER_DBG_XCD_RUNS = 512, // probe only: tiles by workgroup index, runs of 2^(bits 16-19) consecutive tiles per XCD (no ticket)
	SCATTER_XCD_RUN_SHIFT = 16,
	SCATTER_SEG_LEAVES = 1024,  // segmented pass (rsx_hybrid.hpp): the one by the level-2 column
	SCATTER_SEG_SLACK = 2048,   // ... written into per-bucket slots of a scratch array, without counts (SegArgs::slack_cap)
	SCATTER_SELF_PLAN = 1u << 21,   // (bits 12-14 are the column, 16-19 the probe's run length) pass 0 derives the plan itself from the raw counts (SelfPlanArgs, rsx_scatter2.hpp)
	SCATTER_BLIND = 1u << 22,       // segmented pass of a sort WITHOUT a histogram (rsx_hybrid.hpp, rsx_blind_*): runs only while SegCtl::blind says go
	SCATTER_BLIND_TOP = 1u << 23,   // ... its first pass: by the highest column of the plan the sample made, plain tiles, ONE bucket whose 256 digits have a slot each
	// keys-only MSB passes whose buckets go to leaves that sort them anyway (rsx_hybrid.hpp): the order of a bucket's keys does not
	// matter, so the workgroup ranks in ONE row of cells shared by its waves -- no prefix over the waves' rows (the layout phase)
	SCATTER_UNSTABLE = 1u << 24,
	SCATTER_RANK_ASYNC = 1u << 20   // device-scheduled pass of rsx_sort_rank_inplace_async: buffers, index generation and the key-less last pass follow from the plan
};

// Tile shape: NWAVES wavefronts per workgroup, KPT keys per lane => NWAVES*64*KPT keys per tile.
// RANK_: how keys are ranked inside a wavefront
//   RANK_ATOMIC  one returning LDS atomic add per key on the wave's digit counter.  Relies on the LDS
//                handing out the return values of one wave-instruction's same-address lanes in increasing
//                lane order.  That is how gfx950 behaves (tools/ubench/lds_atomic_order.hip: 1.2e10 lane
//                checks, no exception) but it is not documented, so the host verifies it on the device
//                it runs on before selecting this mode (rsx.hip, lds_order_selfcheck).
//   RANK_TABLE   match through per-wave LDS tables (lane masks): no assumption beyond in-order DS issue.
enum { RANK_TABLE = 0, RANK_ATOMIC = 1 };

template <int NWAVES_, int KPT_, int LB_ = 8, int OCC_ = 2, int RANK_ = RANK_ATOMIC> struct TileShape {
	static constexpr int RANK = RANK_;
	static constexpr int NWAVES = NWAVES_;
	static constexpr int KPT = KPT_;
	static constexpr int LB = LB_;     // predecessors' status words fetched per look-back round trip
	static constexpr int OCC = OCC_;   // workgroups per CU the register allocation is bounded for
};
template <typename KT, typename VT, int RANK = RANK_TABLE> struct DefaultShape
	: TileShape<8, ((sizeof(KT) > (size_t)val_bytes<VT>::value ? sizeof(KT) : (size_t)val_bytes<VT>::value) == 8 ? 8 : 16), 8, 2,
	            RANK> {};

template <typename KT, typename VT, typename SH = DefaultShape<KT, VT>> struct ScatterCfg {
	static constexpr int NWAVES = SH::NWAVES;
	static constexpr int BLOCK = NWAVES * 64;
	static constexpr int ELEM = sizeof(KT) > (size_t)val_bytes<VT>::value ? sizeof(KT) : val_bytes<VT>::value;
	static constexpr int KPT = SH::KPT;
	static constexpr int TILE = BLOCK * KPT;
	static constexpr int CHUNK = 16 / ELEM;   // consecutive staged elements one lane writes out together
	static constexpr int HR = 4;              // lane-striped copies of the super-tile histogram
	// RANK_TABLE: the per-wave match tables (256 slots x 16 B) live in the staging area until ranking is done
	static constexpr int TABLE_BYTES = SH::RANK == RANK_TABLE ? NWAVES * 256 * 16 : 0;
	static constexpr int STAGE_BYTES = TILE * ELEM > TABLE_BYTES ? TILE * ELEM : TABLE_BYTES;
	static_assert(NWAVES >= 4, "256 digit threads are needed");
	static_assert(TILE <= 65536, "tile-local positions are packed in 16 bits");
	static_assert(KPT % CHUNK == 0 && KPT % 2 == 0, "whole chunks / rank pairs per lane");
	static_assert(NWAVES * 128 >= 256 * HR, "the super-tile histogram borrows the counter array");
};

// One slot per digit and wave (RANK_TABLE): the lanes of the current round that hold the digit
// (64-bit mask) and how many keys of earlier rounds held it.
struct MatchSlot { u32 mask_lo, mask_hi, count, pad; };

template <typename KT, typename VT, typename ST, typename SH> struct ScatterSmem {
	typedef ScatterCfg<KT, VT, SH> C;
	__attribute__((aligned(16))) unsigned char stage_raw[C::STAGE_BYTES];
	// 16-bit cell per (wave, digit), two digits per word: the wave's digit counter while ranking, then
	// the tile-local start of (wave, digit)'s run.  Phase A borrows it for the super-tile histogram.
	u32 wcell[C::NWAVES][128];
	ST delta[256];               // global offset of a digit's run minus its tile-local offset
	u32 wsum[4];
	u32 ticket;
};

template <typename KT>
__device__ __forceinline__ u32 digit_of(KT raw, const KdfArgs<KT> ka, u32 shift)
{
	u32 d = (u32)(kdf_apply(raw, ka) >> shift) & 0xFFu;
	return d;
}

// 16-byte (or 8-byte) store to an address that is only element-aligned.
template <typename T, int E>
__device__ __forceinline__ void store_chunk(T *dst, const T (&v)[E])
{
	typedef T vec_t __attribute__((ext_vector_type(E)));
	typedef vec_t uvec_t __attribute__((aligned(sizeof(T))));
	vec_t x;
#pragma unroll
	for (int e = 0; e < E; ++e)
		x[e] = v[e];
	*(uvec_t *)dst = x;
}

// Keeps a stream busy for `ticks` of the 100 MHz wall clock (one wave): a known-length occupant for stream / queue
// experiments (multi.py: which stream overlaps RCCL's kernels).
__global__ void rsx_spin_kernel(u64 ticks)
{
	const u64 t0 = wall_clock64();
	while (wall_clock64() - t0 < ticks)
		__builtin_amdgcn_s_sleep(32);
}

#define RSX_STAMP(k)                                                    \
	do {                                                                \
		if (TL && threadIdx.x == 0)                                     \
			tl[(u64)tile * 16 + (k)] = __builtin_readcyclecounter();    \
	} while (0)

// One tile of a super-tile: rank, lay out, stage, write out.  `running` (digit threads) is the
// global offset at which this tile's run of digit tid starts; it is advanced by the tile's count.
template <typename KT, typename VT, typename ST, typename SH, bool FULL, bool TL>
__device__ __forceinline__ void scatter_tile(ScatterSmem<KT, VT, ST, SH> &sm, const KT *__restrict__ kin, KT *__restrict__ kout,
                                             const VT *__restrict__ vin, VT *__restrict__ vout, const u32 tile,
                                             const u64 tile_base, const u32 tile_count, const u32 shift, u64 &running,
                                             const KdfArgs<KT> ka, const u32 flags, u64 *tl)
{
	typedef ScatterCfg<KT, VT, SH> C;
	constexpr int NWAVES = C::NWAVES, BLOCK = C::BLOCK, KPT = C::KPT, CHUNK = C::CHUNK;
	constexpr bool HAS_VAL = val_bytes<VT>::value != 0;
	KT *stage_k = (KT *)sm.stage_raw;
	const u32 tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
	unsigned short *cell16 = (unsigned short *)&sm.wcell[0][0];   // [NWAVES][256] 16-bit cells

	// ---- load: wave w owns [tile_base + w*64*KPT, +64*KPT), lane l of round r the element 64 r + l of it
	const u32 wofs = wid * (64 * KPT) + lane;
	KT key[KPT];
#pragma unroll
	for (int r = 0; r < KPT; ++r) {
		const u32 o = wofs + r * 64;
		key[r] = (FULL || o < tile_count) ? kin[tile_base + o] : (KT)0;
	}

	u32 rkp[KPT / 2];   // 16 bits per key: rank inside the wave, later the tile-local position
#pragma unroll
	for (int i = 0; i < KPT / 2; ++i)
		rkp[i] = 0;
	if constexpr (SH::RANK == RANK_ATOMIC) {
		// ---- rank inside the wave: one returning LDS atomic per key on the wave's own digit counters
		// (16-bit cells, two per word), rounds in memory order, lanes of a round in lane order (see
		// RANK_ATOMIC above).  The rounds do not wait for one another: DS operations of a wave are
		// executed in issue order.
		u32 *wc = sm.wcell[wid];
#pragma unroll
		for (int i = 0; i < 2; ++i)
			wc[lane + 64 * i] = 0;   // the caller's barrier separates this from the previous tile's reads
		if (TL) {
			asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
			RSX_STAMP(2);
		}
#pragma unroll
		for (int r = 0; r < KPT; ++r) {
			const bool valid = FULL || (wofs + r * 64 < tile_count);
			const u32 d = digit_of(key[r], ka, shift);
			if (valid) {
				const u32 sh = (d & 1u) * 16u;
				const u32 old = __hip_atomic_fetch_add(&wc[d >> 1], 1u << sh, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
				rkp[r >> 1] |= ((old >> sh) & 0xFFFFu) << (16 * (r & 1));
			}
		}
	} else {
		// zero this wave's match table (4 KiB = 4 x 16 B per lane); the caller's barrier separates this
		// from the previous tile's reads of the staging area
		{
			u32x4 *t = (u32x4 *)sm.stage_raw + wid * 256 + lane;
#pragma unroll
			for (int i = 0; i < 4; ++i)
				t[i * 64] = u32x4{0u, 0u, 0u, 0u};
		}
		if (TL) {
			asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
			RSX_STAMP(2);
		}
		// ---- rank inside the wave, rounds in memory order.
		// Match through LDS: every lane ORs its lane bit into its digit's slot, reads the slot back
		// (= the lanes of this round with the same digit, and the count of earlier rounds), and the
		// highest of those lanes clears the mask and bumps the count.  DS operations of one wave
		// execute in issue order, so the read sees every lane's OR (and this wave's zeroing above), and
		// the next round's OR sees the cleared mask.
		MatchSlot *tab = (MatchSlot *)sm.stage_raw + wid * 256;
		const u32 lanebit = 1u << (lane & 31);
		const u32 half = lane >> 5;
#pragma unroll
		for (int r = 0; r < KPT; ++r) {
			const bool valid = FULL || (wofs + r * 64 < tile_count);
			const u32 d = digit_of(key[r], ka, shift);
			if (valid) {
				u32 *slot = &tab[d].mask_lo;
				__hip_atomic_fetch_or(slot + half, lanebit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
				RSX_COMPILER_FENCE();
				const u32x4 s = *(const u32x4 *)slot;
				RSX_COMPILER_FENCE();
				const u32 cnt = __popc(s.x) + __popc(s.y);
				const u32 below = __builtin_amdgcn_mbcnt_hi(s.y, __builtin_amdgcn_mbcnt_lo(s.x, 0u));
				if (below == cnt - 1)
					*(u32x4 *)slot = u32x4{0u, 0u, s.z + cnt, 0u};
				RSX_COMPILER_FENCE();
				rkp[r >> 1] |= (s.z + below) << (16 * (r & 1));
			}
		}
	}
	RSX_STAMP(3);
	__syncthreads();
	RSX_STAMP(4);

	// ---- digit thread d: tile count -> tile-local layout and this tile's global offsets
	u32 tile_cnt = 0, incl = 0;
	if (tid < 256) {
#pragma unroll
		for (int w = 0; w < NWAVES; ++w) {
			if constexpr (SH::RANK == RANK_ATOMIC)
				tile_cnt += cell16[w * 256 + tid];
			else
				tile_cnt += ((const MatchSlot *)sm.stage_raw + tid)[w * 256].count;
		}
		incl = tile_cnt;
#pragma unroll
		for (int off = 1; off < 64; off <<= 1) {
			const u32 t = __shfl_up(incl, off);
			if (lane >= (u32)off)
				incl += t;
		}
		if (lane == 63)
			sm.wsum[wid] = incl;
	}
	__syncthreads();
	RSX_STAMP(6);
	if (tid < 256) {
		u32 tbase = 0;
		for (u32 w = 0; w < wid; ++w)
			tbase += sm.wsum[w];
		tbase += incl - tile_cnt;   // tile-local start of digit tid's run
		u32 acc = tbase;
#pragma unroll
		for (int w = 0; w < NWAVES; ++w) {   // counts -> starts, in place (RANK_TABLE: from the tables)
			u32 c;
			if constexpr (SH::RANK == RANK_ATOMIC)
				c = cell16[w * 256 + tid];
			else
				c = ((const MatchSlot *)sm.stage_raw + tid)[w * 256].count;
			cell16[w * 256 + tid] = (unsigned short)acc;
			acc += c;
		}
		sm.delta[tid] = (ST)(running - tbase);   // modulo 2^32 when ST is 32-bit (n < 2^30 then)
		if (TL && (flags & SCATTER_DBG_LINEAR))
			sm.delta[tid] = (ST)tile_base;
		running += tile_cnt;
	}
	__syncthreads();   // RANK_TABLE: every table read is done, the staging area may be overwritten
	RSX_STAMP(7);

	// ---- stage keys in output order
	const unsigned short *wb = cell16 + wid * 256;
#pragma unroll
	for (int r = 0; r < KPT; ++r) {
		const bool valid = FULL || (wofs + r * 64 < tile_count);
		const u32 d = digit_of(key[r], ka, shift);
		const u32 pos = (u32)wb[d] + ((rkp[r >> 1] >> (16 * (r & 1))) & 0xFFFFu);
		if constexpr (HAS_VAL)
			rkp[r >> 1] = (rkp[r >> 1] & ~(0xFFFFu << (16 * (r & 1)))) | (pos << (16 * (r & 1)));
		if (valid)
			stage_k[pos] = key[r];
	}
	RSX_STAMP(8);
	__syncthreads();
	RSX_STAMP(9);

	// ---- write out.  The staged tile is sorted by digit, and consecutive staged elements of one digit
	// go to consecutive addresses: a lane takes CHUNK consecutive elements and, when they share a digit
	// (first == last), stores them with one wide store; chunks straddling a run boundary go element-wise.
	u32 pk[HAS_VAL ? KPT / CHUNK : 1];   // the chunk's digits, one byte each (CHUNK <= 4 whenever there is a payload)
#pragma unroll
	for (int j = 0; j < KPT / CHUNK; ++j) {
		if (j % 4 == 0)
			__builtin_amdgcn_sched_barrier(0);   // keep at most four chunks' registers alive at a time
		const u32 i0 = CHUNK * (tid + j * BLOCK);
		KT kv[CHUNK];
		u32 d[CHUNK];
		{
			typedef KT kvec_t __attribute__((ext_vector_type(CHUNK)));
			const kvec_t x = *(const kvec_t *)(stage_k + i0);
#pragma unroll
			for (int e = 0; e < CHUNK; ++e)
				kv[e] = x[e];
		}
#pragma unroll
		for (int e = 0; e < CHUNK; ++e)
			d[e] = digit_of(kv[e], ka, shift);
		if constexpr (HAS_VAL) {
			u32 p = 0;
#pragma unroll
			for (int e = 0; e < CHUNK; ++e)
				p |= d[e] << (8 * e);
			pk[j] = p;
		}
		if (!(flags & SCATTER_SKIP_KEYS) && !(TL && (flags & SCATTER_DBG_NOSTORE))) {
			const bool whole = FULL || i0 + CHUNK <= tile_count;
			if (sizeof(KT) >= 4 && whole && d[0] == d[CHUNK - 1]) {
				store_chunk<KT, CHUNK>(kout + (ST)(sm.delta[d[0]] + i0), kv);
			} else {
#pragma unroll
				for (int e = 0; e < CHUNK; ++e)
					if (FULL || i0 + e < tile_count)
						kout[(ST)(sm.delta[d[e]] + i0 + e)] = kv[e];
			}
		}
	}
	if constexpr (HAS_VAL) {
		// payloads: same positions, through the same staging area
		VT *stage_v = (VT *)sm.stage_raw;
		VT val[KPT];
		if (flags & SCATTER_GEN_INDEX) {
#pragma unroll
			for (int r = 0; r < KPT; ++r)
				val[r] = (VT)(tile_base + wofs + r * 64);
		} else {
#pragma unroll
			for (int r = 0; r < KPT; ++r) {
				const u32 o = wofs + r * 64;
				val[r] = (FULL || o < tile_count) ? vin[tile_base + o] : (VT)0;
			}
		}
		__syncthreads();
#pragma unroll
		for (int r = 0; r < KPT; ++r) {
			const bool valid = FULL || (wofs + r * 64 < tile_count);
			if (valid)
				stage_v[(rkp[r >> 1] >> (16 * (r & 1))) & 0xFFFFu] = val[r];
		}
		__syncthreads();
#pragma unroll
		for (int j = 0; j < KPT / CHUNK; ++j) {
			const u32 i0 = CHUNK * (tid + j * BLOCK);
			VT vv[CHUNK];
			{
				typedef VT vvec_t __attribute__((ext_vector_type(CHUNK)));
				const vvec_t x = *(const vvec_t *)(stage_v + i0);
#pragma unroll
				for (int e = 0; e < CHUNK; ++e)
					vv[e] = x[e];
			}
			const u32 d0 = pk[j] & 0xFFu, dl = (pk[j] >> (8 * (CHUNK - 1))) & 0xFFu;
			const bool whole = FULL || i0 + CHUNK <= tile_count;
			if (whole && d0 == dl) {
				store_chunk<VT, CHUNK>(vout + (ST)(sm.delta[d0] + i0), vv);
			} else {
#pragma unroll
				for (int e = 0; e < CHUNK; ++e)
					if (FULL || i0 + e < tile_count)
						vout[(ST)(sm.delta[(pk[j] >> (8 * e)) & 0xFFu] + i0 + e)] = vv[e];
			}
		}
	}
	if (TL) {
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		RSX_STAMP(10);
	}
}

// A workgroup takes a SUPER-TILE of `tps` consecutive tiles (one ticket, one status row, one
// look-back) and then scatters its tiles one after the other.  Chaining tile by tile does not scale
// on this chip: a status word takes about 1.3 us to travel under load, during which dozens of tiles
// reach their look-back, so every tile would have to read dozens of predecessors.  With a super-tile
// the chain advances tps times slower and its latency is paid once per tps tiles.
//   phase A  count the super-tile's digits (one extra read of its keys; the second read, in phase B,
//            comes out of L2 / Infinity Cache), publish the aggregate, look back, publish the prefix;
//   phase B  rank / stage / write each tile with offsets carried from tile to tile in registers.
// grid = number of super-tiles.  gbase[digit] is the exclusive offset of the digit for this pass.
template <typename KT, typename VT, typename ST, typename SH = DefaultShape<KT, VT>, bool TL = false>
__global__ __launch_bounds__(SH::NWAVES * 64, (SH::NWAVES * SH::OCC + 3) / 4) void rsx_scatter_kernel(
	const KT *__restrict__ kin, KT *__restrict__ kout, const VT *__restrict__ vin, VT *__restrict__ vout, u64 n, u32 shift,
	const u64 *__restrict__ gbase, u32 tps, ST *status, u32 *ticket, KdfArgs<KT> ka, u32 flags, u64 *tl)
{
	typedef ScatterCfg<KT, VT, SH> C;
	typedef StatusBits<ST> SB;
	constexpr int BLOCK = C::BLOCK, KPT = C::KPT, HR = C::HR;
	__shared__ ScatterSmem<KT, VT, ST, SH> sm;
	const u32 tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
	const u64 t_start = TL ? __builtin_readcyclecounter() : 0;
	if (tid == 0)
		sm.ticket = atomicAdd(ticket, 1u);   // super-tiles are handed out in start order => look-back cannot deadlock
	u32 *hist = &sm.wcell[0][0];             // [256][HR] while phase A runs
	for (u32 i = tid; i < 256 * HR; i += BLOCK)
		hist[i] = 0;
	__syncthreads();
	const u32 stile = __builtin_amdgcn_readfirstlane(sm.ticket);   // wave-uniform: addresses below stay scalar
	const u64 first_tile = (u64)stile * tps;
	const u64 beg = first_tile * C::TILE;
	u64 end = beg + (u64)tps * C::TILE;
	if (end > n)
		end = n;

	// ---- phase A: digit counts of the whole super-tile (order does not matter here)
	{
		constexpr int PA = KPT < 16 ? KPT : 16;   // loads in flight per lane
		const u32 wofs = wid * (64 * KPT) + lane;
		for (u64 base = beg; base < end; base += C::TILE) {
			const u32 cnt = (end - base) < (u64)C::TILE ? (u32)(end - base) : (u32)C::TILE;
#pragma unroll 1
			for (int r0 = 0; r0 < KPT; r0 += PA) {
				KT cur[PA];
#pragma unroll
				for (int r = 0; r < PA; ++r) {
					const u32 o = wofs + (r0 + r) * 64;
					cur[r] = o < cnt ? kin[base + o] : (KT)0;
				}
#pragma unroll
				for (int r = 0; r < PA; ++r) {
					const u32 o = wofs + (r0 + r) * 64;
					if (o < cnt)
						atomicAdd(&hist[digit_of(cur[r], ka, shift) * HR + (lane & (HR - 1))], 1u);
				}
			}
		}
	}
	__syncthreads();

	// ---- digit thread d: publish the aggregate, look back along the chain of super-tiles, publish the prefix
	u64 running = 0;
	if (TL && tid == 0)
		tl[first_tile * 16 + 13] = __builtin_readcyclecounter();
	if (tid < 256) {
		u32 st_cnt = 0;
#pragma unroll
		for (int r = 0; r < HR; ++r)
			st_cnt += hist[tid * HR + r];
		ST *my_status = status + (u64)stile * 256 + tid;
		const ST word = ((ST)(stile == 0 ? ST_PREFIX : ST_AGGREGATE) << SB::SHIFT) | (ST)st_cnt;
		__hip_atomic_store(my_status, word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		u64 excl = 0;
		u32 depth = 0;
		if (stile != 0) {
			// LB predecessors are fetched per round trip (independent loads), then consumed in order:
			// aggregates are summed until the first inclusive prefix; an empty word ends the batch.
			constexpr int LB = SH::LB;
			long back = (long)stile - 1;   // nearest predecessor not consumed yet
			const ST *col = status + tid;
			for (;;) {
				ST w[LB];
#pragma unroll
				for (int j = 0; j < LB; ++j) {
					const long t = back - j > 0 ? back - j : 0;   // super-tile 0 always holds a prefix: safe filler
					w[j] = __hip_atomic_load(col + (u64)t * 256, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				}
				bool done = false;
				int used = 0;
#pragma unroll
				for (int j = 0; j < LB; ++j) {
					const u32 f = (u32)(w[j] >> SB::SHIFT);
					if (!done && used == j && f != ST_EMPTY) {
						excl += (u64)(w[j] & SB::VALMASK);
						++used;
						++depth;
						done = f == ST_PREFIX;
					}
				}
				if (done)
					break;
				back -= used;
				if (used == 0)
					__builtin_amdgcn_s_sleep(1);
			}
			const ST pword = ((ST)ST_PREFIX << SB::SHIFT) | (ST)(excl + st_cnt);
			__hip_atomic_store(my_status, pword, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		}
		running = gbase[tid] + excl;
		if (TL && tid == 0) {
			tl[first_tile * 16 + 0] = t_start;
			tl[first_tile * 16 + 1] = __builtin_readcyclecounter();
			tl[first_tile * 16 + 12] = depth;
		}
	}
	__syncthreads();   // the histogram (aliasing wcell) has been consumed

	// ---- phase B: the tiles, in order
	for (u64 base = beg; base < end; base += C::TILE) {
		const u32 tile = (u32)(base / C::TILE);
		const u32 tile_count = (end - base) < (u64)C::TILE ? (u32)(end - base) : (u32)C::TILE;
		if (tile_count == (u32)C::TILE)
			scatter_tile<KT, VT, ST, SH, true, TL>(sm, kin, kout, vin, vout, tile, base, tile_count, shift, running, ka, flags,
			                                       tl);
		else
			scatter_tile<KT, VT, ST, SH, false, TL>(sm, kin, kout, vin, vout, tile, base, tile_count, shift, running, ka, flags,
			                                        tl);
		__syncthreads();   // staging reads done before the next tile's counters / tables are reset
	}
}

// =============================================================================
// Self-check for RANK_ATOMIC / rsx_scatter2_kernel
// =============================================================================
// Do returning LDS atomics hand out their return values in increasing lane order when several lanes of
// one wave-instruction hit the same address, and in issue order across instructions?  Every wave runs
// rounds of `old = atomicAdd(&cnt[d], 1)` on digits with heavy collisions and compares `old` with the
// rank obtained independently from ballots (the 8-ballot match).  *bad counts the disagreements.
__global__ __launch_bounds__(512) void rsx_lds_order_check_kernel(u64 *bad, u32 seed, int rounds)
{
	__shared__ u32 cnt[8][256];
	__shared__ u32 ref[8][256];
	const u32 tid = threadIdx.x, wid = tid >> 6;
	for (u32 i = tid; i < 8 * 256; i += 512) {
		(&cnt[0][0])[i] = 0;
		(&ref[0][0])[i] = 0;
	}
	__syncthreads();
	u32 x = seed ^ (blockIdx.x * 2654435761u) ^ (tid * 40503u);
	u64 nbad = 0;
	for (int r = 0; r < rounds; ++r) {
		x = x * 1664525u + 1013904223u;
		u32 d;
		switch ((blockIdx.x + r / 64) & 3) {
		case 0: d = (x >> 13) & 0xFF; break;                                  // uniform
		case 1: d = (x >> 13) & 1; break;                                     // two digits
		case 2: d = ((x >> 13) & 7) * 32; break;                              // eight addresses on one bank
		default: d = ((x >> 9) % 3 == 0) ? 200 : ((x >> 13) & 0xFF); break;   // one hot digit + uniform
		}
		const u32 old = __hip_atomic_fetch_add(&cnt[wid][d], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
		u64 m = ~0ull;
#pragma unroll
		for (int b = 0; b < 8; ++b) {
			const bool bit = (d >> b) & 1u;
			const u64 bal = __ballot(bit);
			m &= bit ? bal : ~bal;
		}
		const u32 below = mbcnt64(m);
		const u32 prev = ref[wid][d];
		RSX_COMPILER_FENCE();
		if (below == (u32)__popcll(m) - 1)
			ref[wid][d] = prev + (u32)__popcll(m);
		RSX_COMPILER_FENCE();
		if (old != prev + below)
			++nbad;
	}
	if (nbad)
		atomicAdd(bad, nbad);
}

// The same question in the PRODUCTION geometry of rsx_scatter2_kernel (ADVICE / VERDICT round 1: the check above runs 8 waves
// that issue nothing but atomics): 16 waves per workgroup, 32-bit (wave, digit) cells that start at run offsets, eight
// returning atomics in flight per wave before their values are used, and the staging traffic of the real pass between
// them -- 4-byte stores at the returned positions and 16-byte stores / reads of the wave's scratch row -- all in a
// 96 KiB shared block, one workgroup per CU.  A lane's returned value must be (the cell before the instruction)
// + (the number of lower lanes with the same digit), the cell before the instruction being what the ballots of all
// earlier rounds add up to.
struct LdsOrderSmem {
	__attribute__((aligned(16))) u32 stage[16384];
	u32 cell[16][256];
	u32 ref[16][256];
};
__global__ __launch_bounds__(1024) void rsx_lds_order_check2_kernel(u64 *bad, u32 seed, int rounds)
{
	__shared__ LdsOrderSmem sm;
	const u32 tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
	for (u32 i = tid; i < 16 * 256; i += 1024) {
		const u32 start = (i * 2654435761u) >> 18;              // any run start below 16384
		(&sm.cell[0][0])[i] = start;
		(&sm.ref[0][0])[i] = start;
	}
	__syncthreads();
	u32 x = seed ^ (blockIdx.x * 2654435761u) ^ (tid * 40503u);
	u64 nbad = 0;
	u32 *wc = sm.cell[wid];
	u32x4 *row = (u32x4 *)sm.stage + (u32)wid * 128 + 64;       // (a scratch row of the wave inside the staging area)
	for (int r0 = 0; r0 < rounds; r0 += 8) {
		u32 d[8], old[8];
#pragma unroll
		for (int r = 0; r < 8; ++r) {
			x = x * 1664525u + 1013904223u;
			switch ((blockIdx.x + (r0 + r) / 64) & 3) {
			case 0: d[r] = (x >> 13) & 0xFF; break;                                  // uniform
			case 1: d[r] = (x >> 13) & 3; break;                                     // four digits
			case 2: d[r] = ((x >> 13) & 7) * 32; break;                              // eight addresses on one bank
			default: d[r] = ((x >> 9) % 3 == 0) ? 200 : ((x >> 13) & 0xFF); break;   // one hot digit + uniform
			}
		}
		// eight atomics in flight, as rsx_scatter2_kernel's stage_batch issues them
#pragma unroll
		for (int r = 0; r < 8; ++r)
			old[r] = __hip_atomic_fetch_add(&wc[d[r]], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
		// the traffic of the real pass around them: keys stored at the returned positions, a 16-byte row written and read
#pragma unroll
		for (int r = 0; r < 8; ++r)
			sm.stage[old[r] & 16383u] = x + r;
		row[lane & 63] = u32x4{x, d[0], old[0], (u32)r0};
		RSX_COMPILER_FENCE();
		const u32x4 back = row[(lane + 1) & 63];
		x ^= back[0] & 1u;
#pragma unroll
		for (int r = 0; r < 8; ++r) {
			u64 m = ~0ull;
#pragma unroll
			for (int b = 0; b < 8; ++b) {
				const bool bit = (d[r] >> b) & 1u;
				const u64 bal = __ballot(bit);
				m &= bit ? bal : ~bal;
			}
			const u32 below = mbcnt64(m);
			const u32 prev = sm.ref[wid][d[r]];
			RSX_COMPILER_FENCE();
			if (below == (u32)__popcll(m) - 1)
				sm.ref[wid][d[r]] = prev + (u32)__popcll(m);
			RSX_COMPILER_FENCE();
			if (old[r] != prev + below)
				++nbad;
		}
	}
	if (nbad)
		atomicAdd(bad, nbad);
}

// RSX_VERIFY=1: one tile of a finished scatter pass, re-ranked WITHOUT LDS atomics (ballot match, one wave, memory order)
// and compared with what the pass wrote.  The tile's keys with digit d must stand, in input order, at
// gbase[d] + (keys of digit d in earlier tiles) + 0, 1, 2, ...; the middle term is the inclusive prefix the predecessor
// tile left in its status word (every tile ends as ST_PREFIX).  `kout` is checked as `out_bytes`-wide values of
// kdf(key) >> oshift when the pass narrowed its keys, as the key itself otherwise; payloads (val_bytes 0 / 4 / 8; vin ==
// nullptr: the element's index) likewise.  *bad counts the mismatches.
template <typename KT, typename ST>
__global__ __launch_bounds__(64) void rsx_verify_tile_kernel(const KT *__restrict__ kin, const void *__restrict__ kout,
                                                             const void *__restrict__ vin, const void *__restrict__ vout, u64 n,
                                                             u32 shift, const u64 *__restrict__ gbase, const ST *__restrict__ status,
                                                             u32 tile, u32 tile_elems, KdfArgs<KT> ka, u32 out_bytes, u32 oshift,
                                                             u32 val_bytes, u32 skip_keys, u64 *bad, u32 inject = 0,
                                                             const Plan *__restrict__ dplan = nullptr, u32 pass_index = 0,
                                                             u32 async_kind = 0, const void *__restrict__ kalt = nullptr)
{
	// (inject: test hook, XORed into every expected key: the mismatch path end to end)
	typedef StatusBits<ST> SB_;
	// A device-scheduled pass (async_kind 1: rsx_sort[_pairs]_inplace_async, 2: rsx_sort_rank_inplace_async): column and
	// buffers follow from the device-side plan exactly as in rsx_scatter2_kernel; a pass that did not run is not checked.
	if (dplan) {
		if (dplan->sorted || pass_index >= dplan->ncols || dplan->hyb)
			return;
		const u32 col = dplan->cols[pass_index];
		shift = 8 * col;
		gbase += 256 * col;
		if (async_kind == 2) {
			const u32 P = dplan->ncols, i = pass_index;
			const KT *k0 = (const KT *)kout, *k1 = (const KT *)kalt;
			kin = i == 0 ? kin : (((i - 1) & 1u) ? k1 : k0);
			kout = (i & 1u) ? (const void *)k1 : (const void *)k0;
			const u32 w = (P - 1 - i) & 1u;
			const void *h0 = vin, *h1 = vout;
			vout = w ? h1 : h0;
			vin = i == 0 ? nullptr : (w ? h0 : h1);
			skip_keys = i == P - 1;
		} else if (pass_index & 1u) {
			const KT *t = kin;
			kin = (const KT *)kout;
			kout = t;
			const void *u = vin;
			vin = vout;
			vout = u;
		}
	}
	__shared__ u64 cursor[256];
	const u32 lane = threadIdx.x;
	for (u32 d = lane; d < 256; d += 64) {
		u64 before = 0;
		if (tile != 0) {
			const ST w = __hip_atomic_load(status + ((u64)(tile - 1) * 256 + d), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			before = (u64)(w & SB_::VALMASK);
			if ((u32)(w >> SB_::SHIFT) != ST_PREFIX)
				atomicAdd(bad, 1ull << 32);   // (the chain itself is broken)
		}
		cursor[d] = gbase[d] + before;
	}
	__syncthreads();
	const u64 base = (u64)tile * tile_elems;
	const u64 end = base + tile_elems < n ? base + tile_elems : n;
	u64 nbad = 0;
	for (u64 i0 = base; i0 < end; i0 += 64) {
		const u64 i = i0 + lane;
		const bool have = i < end;
		const KT key = have ? kin[i] : (KT)0;
		const KT kk = kdf_apply(key, ka);
		const u32 d = have ? (u32)(kk >> shift) & 0xFFu : 0x100u;
		u64 m = __ballot(have);
#pragma unroll
		for (int b = 0; b < 8; ++b) {
			const bool bit = (d >> b) & 1u;
			const u64 bal = __ballot(bit);
			m &= bit ? bal : ~bal;
		}
		if (have) {
			const u64 at = cursor[d & 0xFFu] + mbcnt64(m);
			if (!skip_keys) {
				u64 want = (oshift || out_bytes != sizeof(KT) ? (u64)(kk >> oshift) : (u64)key) ^ inject, got = 0;
				if (out_bytes < 8)
					want &= (1ull << (8 * out_bytes)) - 1;
				switch (out_bytes) {
				case 1: got = ((const uint8_t *)kout)[at]; break;
				case 2: got = ((const uint16_t *)kout)[at]; break;
				case 4: got = ((const u32 *)kout)[at]; break;
				default: got = ((const u64 *)kout)[at]; break;
				}
				nbad += got != want;
			}
			if (val_bytes == 4)
				nbad += ((const u32 *)vout)[at] != (vin ? ((const u32 *)vin)[i] : (u32)i);
			else if (val_bytes == 8)
				nbad += ((const u64 *)vout)[at] != (vin ? ((const u64 *)vin)[i] : (u64)i);
		}
		__syncthreads();
		if (have && mbcnt64(m) == (u32)__popcll(m) - 1)   // the highest lane of each digit group
			cursor[d & 0xFFu] += (u64)__popcll(m);
		__syncthreads();
	}
	if (nbad)
		atomicAdd(bad, nbad);
}

// README.md:716-758 "key compaction": dst[i] = the varying bits of kdf(src[i]) packed together (a software bit extract
// over at most 8 runs of contiguous mask bits).  All other bits are the same in every key, so the packed values order
// exactly as the keys do; a rank sort of them needs ceil(varying bits / 8) passes instead of one per varying byte.
struct BitRuns {
	uint8_t n;            // runs
	uint8_t src[8];       // first bit of the run in the key
	uint8_t len[8];       // bits
	uint8_t dst[8];       // first bit of the run in the packed value
};
template <typename KT, typename OT>
__global__ __launch_bounds__(256) void rsx_compact_bits_kernel(const KT *__restrict__ src, OT *__restrict__ dst, u64 n, KdfArgs<KT> ka,
                                                               BitRuns runs)
{
	for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < n; i += (u64)gridDim.x * 256) {
		const u64 k = (u64)kdf_apply(src[i], ka);
		u64 o = 0;
#pragma unroll
		for (int r = 0; r < 8; ++r)
			if (r < runs.n)
				o |= ((k >> runs.src[r]) & ((1ull << runs.len[r]) - 1)) << runs.dst[r];
		dst[i] = (OT)o;
	}
}

// =============================================================================
// Small helpers
// =============================================================================

// splitmix64 evaluated counter-based: output number k (0-based) of the stream seeded with s
// is mix(s + (k+1) * 0x9E3779B97F4A7C15) (SURVEY.md section 4 generator).
template <typename T>
__global__ void rsx_fill_splitmix_kernel(T *dst, u64 n, u64 seed, u64 mask, u64 first)
{
	const u64 stride = (u64)gridDim.x * blockDim.x;
	for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
		u64 z = seed + (first + i + 1) * 0x9E3779B97F4A7C15ull;
		z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
		z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
		z ^= z >> 31;
		dst[i] = (T)(z & mask);
	}
}

// dst = 0 .. n-1 if the device-side plan says "sorted": what rs_sort_rank leaves in the first half then
// (radix_sort_rank.hpp:52,:55-57); the last step of rsx_sort_rank_inplace_async.
template <typename IT>
__global__ void rsx_iota_if_sorted_kernel(IT *dst, u64 n, const Plan *__restrict__ plan)
{
	if (!plan->sorted)
		return;
	for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x)
		dst[i] = (IT)i;
}

// RSX_VERIFY=2: the whole-result check of a keys-only sort.  out[0] += the descents of the derived keys (kdf(a[i]) > kdf(a[i+1])),
// out[1] += their sum, out[2] ^= a mix of them: a sorted array with the input's sum and mix is the sorted input (keys-only:
// equal keys are indistinguishable, so this is the whole contract) -- whichever route the sort took, leaves included.
template <typename KT>
__global__ __launch_bounds__(256) void rsx_checksum_kernel(const KT *__restrict__ a, u64 n, KdfArgs<KT> ka, u64 *out)
{
	u64 bad = 0, sum = 0, mix = 0;
	for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) {
		const KT k = kdf_apply(a[i], ka);
		if (i + 1 < n && k > kdf_apply(a[i + 1], ka))
			++bad;
		sum += (u64)k;
		mix ^= ((u64)k + 0x9E3779B97F4A7C15ull) * 0xBF58476D1CE4E5B9ull;
	}
	__shared__ u64 s[3];
	if (threadIdx.x < 3)
		s[threadIdx.x] = 0;
	__syncthreads();
	atomicAdd((unsigned long long *)&s[0], (unsigned long long)bad);
	atomicAdd((unsigned long long *)&s[1], (unsigned long long)sum);
	atomicXor((unsigned long long *)&s[2], (unsigned long long)mix);
	__syncthreads();
	if (threadIdx.x == 0) {
		atomicAdd((unsigned long long *)&out[0], (unsigned long long)s[0]);
		atomicAdd((unsigned long long *)&out[1], (unsigned long long)s[1]);
		atomicXor((unsigned long long *)&out[2], (unsigned long long)s[2]);
	}
}

// RSX_VERIFY=2 for key + payload sorts: out[0] += descents of the keys, out[1] += sum of the derived keys, out[2] ^= a mix of
// every (key, payload) PAIR -- a result with no descent and the input's sum and pair mix is a sorted permutation of the
// input's pairs (which of two equal keys' payloads comes first -- stability -- is not visible to it).
template <typename KT, typename VT>
__global__ __launch_bounds__(256) void rsx_checksum_pairs_kernel(const KT *__restrict__ k, const VT *__restrict__ v, u64 n, KdfArgs<KT> ka,
                                                                 u64 *out)
{
	u64 bad = 0, sum = 0, mix = 0;
	for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) {
		const KT x = kdf_apply(k[i], ka);
		if (i + 1 < n && x > kdf_apply(k[i + 1], ka))
			++bad;
		sum += (u64)x;
		mix ^= (((u64)x + 0x9E3779B97F4A7C15ull) * 0xBF58476D1CE4E5B9ull) ^ (((u64)v[i] + 0x632BE59BD9B4E019ull) * 0x94D049BB133111EBull + (u64)x);
	}
	__shared__ u64 s[3];
	if (threadIdx.x < 3)
		s[threadIdx.x] = 0;
	__syncthreads();
	atomicAdd((unsigned long long *)&s[0], (unsigned long long)bad);
	atomicAdd((unsigned long long *)&s[1], (unsigned long long)sum);
	atomicXor((unsigned long long *)&s[2], (unsigned long long)mix);
	__syncthreads();
	if (threadIdx.x == 0) {
		atomicAdd((unsigned long long *)&out[0], (unsigned long long)s[0]);
		atomicAdd((unsigned long long *)&out[1], (unsigned long long)s[1]);
		atomicXor((unsigned long long *)&out[2], (unsigned long long)s[2]);
	}
}

// ... for rank sorts (radix_sort_rank.hpp:97-112): out[0] += places where keys[r[i]] > keys[r[i + 1]], or the keys are equal and
// r[i] > r[i + 1] (not stable), or r[i] >= n; out[1] += sum of the ranks, out[2] ^= a mix of them.  keys == nullptr: only the
// sum and the mix of 0 .. n-1 (what a permutation must give).
template <typename KT, typename IT>
__global__ __launch_bounds__(256) void rsx_check_ranks_kernel(const KT *__restrict__ keys, const IT *__restrict__ r, u64 n, KdfArgs<KT> ka,
                                                              u64 *out)
{
	u64 bad = 0, sum = 0, mix = 0;
	for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) {
		const u64 a = keys ? (u64)r[i] : i;
		if (keys) {
			if (a >= n) {
				++bad;
			} else if (i + 1 < n) {
				const u64 b = (u64)r[i + 1];
				if (b < n) {
					const KT ka_ = kdf_apply(keys[a], ka), kb_ = kdf_apply(keys[b], ka);
					if (ka_ > kb_ || (ka_ == kb_ && a > b))
						++bad;
				}
			}
		}
		sum += a;
		mix ^= (a + 0x9E3779B97F4A7C15ull) * 0xBF58476D1CE4E5B9ull;
	}
	__shared__ u64 s[3];
	if (threadIdx.x < 3)
		s[threadIdx.x] = 0;
	__syncthreads();
	atomicAdd((unsigned long long *)&s[0], (unsigned long long)bad);
	atomicAdd((unsigned long long *)&s[1], (unsigned long long)sum);
	atomicXor((unsigned long long *)&s[2], (unsigned long long)mix);
	__syncthreads();
	if (threadIdx.x == 0) {
		atomicAdd((unsigned long long *)&out[0], (unsigned long long)s[0]);
		atomicAdd((unsigned long long *)&out[1], (unsigned long long)s[1]);
		atomicXor((unsigned long long *)&out[2], (unsigned long long)s[2]);
	}
}

template <typename IT>
__global__ void rsx_iota_kernel(IT *dst, u64 n)
{
	const u64 stride = (u64)gridDim.x * blockDim.x;
	for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
		dst[i] = (IT)i;
}

// Keys only, ONE kept column (always the case for 1-byte keys): every other byte of kdf(key) is the same in all keys
// (radix_sort.hpp:64-70: that is what skipping a column means) and the KDF is a bijection on bit patterns, so the sorted
// array IS the histogram -- count[d] copies of the key whose kept byte is d, for d ascending.  No scatter: `out` is written
// from the kept column's exclusive offsets.  Does nothing unless the device-side plan says "not sorted, one column".
template <typename KT> __device__ __forceinline__ KT kdf_invert(KT y, const KdfArgs<KT> a)
{
	typedef typename std::make_signed<KT>::type ST;
	y = (KT)(y ^ a.desc);
	const KT nsign = (KT)~(KT)((ST)y >> (sizeof(KT) * 8 - 1));   // all ones iff the top bit of the derived key is clear
	return (KT)(y ^ ((nsign & a.fmask) | a.sflip));
}

template <typename KT>
__global__ __launch_bounds__(256) void rsx_fill_runs_kernel(KT *__restrict__ out, u64 n, const u64 *__restrict__ ghist,
                                                            const KT *__restrict__ src, KdfArgs<KT> ka, const Plan *__restrict__ plan)
{
	if (plan->sorted || plan->ncols != 1)
		return;
	const u32 col = plan->cols[0];
	__shared__ u64 offs[257];
	const u32 tid = threadIdx.x;
	offs[tid] = ghist[256 * col + tid];
	if (tid == 0)
		offs[256] = n;
	__syncthreads();
	const KT k0 = (KT)(kdf_apply(src[0], ka) & (KT)~((KT)0xFF << (8 * col)));
	constexpr u32 V = 16 / sizeof(KT);
	typedef KT kvec_t __attribute__((ext_vector_type(V)));
	const u64 nvec = n / V, stride = (u64)gridDim.x * blockDim.x;
	// the digit of output position i: the last d with offs[d] <= i
	auto digit_at = [&](u64 i) {
		u32 lo = 0, hi = 256;   // offs[lo] <= i < offs[hi]
#pragma unroll
		for (int s = 0; s < 8; ++s) {
			const u32 mid = (lo + hi) >> 1;
			if (offs[mid] <= i)
				lo = mid;
			else
				hi = mid;
		}
		return lo;
	};
	for (u64 v = (u64)blockIdx.x * blockDim.x + tid; v < nvec; v += stride) {
		const u64 i0 = v * V;
		u32 d = digit_at(i0);
		kvec_t x;
		if (offs[d + 1] >= i0 + V) {   // the whole vector inside one run
			const KT k = kdf_invert<KT>((KT)(k0 | ((KT)d << (8 * col))), ka);
#pragma unroll
			for (u32 e = 0; e < V; ++e)
				x[e] = k;
		} else {
#pragma unroll
			for (u32 e = 0; e < V; ++e) {
				while (offs[d + 1] <= i0 + e)
					++d;
				x[e] = kdf_invert<KT>((KT)(k0 | ((KT)d << (8 * col))), ka);
			}
		}
		*(kvec_t *)(out + i0) = x;
	}
	if (blockIdx.x == 0 && tid < (u32)(n - nvec * V)) {
		const u64 i = nvec * V + tid;
		out[i] = kdf_invert<KT>((KT)(k0 | ((KT)digit_at(i) << (8 * col))), ka);
	}
}

// ---- 2-byte keys, keys only, both columns kept: ONE 16-bit digit (README.md:781-811, "wider digits") -- the sorted array is the
// joint histogram of the two bytes written out, so the keys are read (twice, see below) and written once instead of
// read once and read + scattered twice.  The three kernels do nothing unless the device-side plan says "not sorted, two columns".
//
// rsx_joint16_kernel: 65536 bins do not fit the LDS as 32-bit counters, so a workgroup counts the half of the bins its
// parity names (derived keys with the top bit clear / set) in 128 KiB and skips the other keys; every key is read by one
// workgroup of either kind.  The counts are added to the global table once per workgroup.
__global__ __launch_bounds__(1024) void rsx_joint16_kernel(const uint16_t *__restrict__ src, u64 n, KdfArgs<uint16_t> ka,
                                                           u32 *__restrict__ joint, const Plan *__restrict__ plan)
{
	if (plan->sorted || plan->ncols != 2)
		return;
	__shared__ u32 tab[32768];
	// workgroups b and b + 8 (the same XCD: workgroups are dealt to the 8 XCDs in turn) read the same keys, one for each half
	// of the bins, so that the second read finds them in that XCD's L2
	const u32 tid = threadIdx.x, half = (blockIdx.x >> 3) & 1u, slot = (blockIdx.x & 7u) | ((blockIdx.x >> 4) << 3),
	          nslots = gridDim.x >> 1;
	for (u32 i = tid; i < 32768u; i += 1024u)
		tab[i] = 0;
	__syncthreads();
	typedef uint16_t kvec_t __attribute__((ext_vector_type(8)));
	const u64 nvec = n / 8;
	const bool aligned = (((uintptr_t)src) & 15) == 0;
	auto count = [&](const uint16_t raw) {
		const u32 k = kdf_apply(raw, ka);
		if ((k >> 15) == half)
			atomicAdd(&tab[k & 0x7FFFu], 1u);
	};
	if (aligned) {
		for (u64 v = (u64)slot * 1024u + tid; v < nvec; v += (u64)nslots * 1024u) {
			const kvec_t x = *(const kvec_t *)(src + v * 8);
#pragma unroll
			for (int e = 0; e < 8; ++e)
				count(x[e]);
		}
		for (u64 i = nvec * 8 + (u64)slot * 1024u + tid; i < n; i += (u64)nslots * 1024u)
			count(src[i]);
	} else {
		for (u64 i = (u64)slot * 1024u + tid; i < n; i += (u64)nslots * 1024u)
			count(src[i]);
	}
	__syncthreads();
	for (u32 i = tid; i < 32768u; i += 1024u)
		if (tab[i])
			atomicAdd(&joint[half * 32768u + i], tab[i]);
}

// joint[65536] counts -> offs[65537] exclusive offsets (one workgroup: 64 bins per thread, a scan of the 1024 sums; 36 us --
// a version with coalesced loads and in-wave scans of 64 chunks per thread needs 192 registers and takes 99)
__global__ __launch_bounds__(1024) void rsx_joint16_scan_kernel(const u32 *__restrict__ joint, u64 *__restrict__ offs, u64 n,
                                                                const Plan *__restrict__ plan)
{
	if (plan->sorted || plan->ncols != 2)
		return;
	__shared__ u64 part[1024];
	const u32 tid = threadIdx.x;
	const u32x4 *j4 = (const u32x4 *)joint + tid * 16;   // this thread's 64 bins: sixteen 16-byte loads, twice
	u64 sum = 0;
#pragma unroll 4
	for (u32 j = 0; j < 16; ++j) {
		const u32x4 v = j4[j];
		sum += (u64)v.x + v.y + v.z + v.w;
	}
	part[tid] = sum;
	__syncthreads();
	for (u32 off = 1; off < 1024; off <<= 1) {
		const u64 add = tid >= off ? part[tid - off] : 0;
		__syncthreads();
		part[tid] += add;
		__syncthreads();
	}
	u64 run = part[tid] - sum;
	typedef u64 u64x2 __attribute__((ext_vector_type(2)));
	u64x2 *o2 = (u64x2 *)(offs + tid * 64);
#pragma unroll 4
	for (u32 j = 0; j < 16; ++j) {
		const u32x4 v = j4[j];
		u64x2 a, c;
		a.x = run;
		a.y = run + v.x;
		c.x = a.y + v.y;
		c.y = c.x + v.z;
		run = c.y + v.w;
		o2[2 * j] = a;
		o2[2 * j + 1] = c;
	}
	if (tid == 1023)
		offs[65536] = n;
}

// out[i] = the key whose derived value is the last bin b with offs[b] <= i.  A workgroup writes a contiguous piece of the
// output; the bins that piece spans (found by two searches of the global table) are searched in the LDS if they fit.
__global__ __launch_bounds__(256) void rsx_fill16_kernel(uint16_t *__restrict__ out, u64 n, const u64 *__restrict__ offs,
                                                         KdfArgs<uint16_t> ka, const Plan *__restrict__ plan)
{
	if (plan->sorted || plan->ncols != 2)
		return;
	typedef uint16_t kvec_t __attribute__((ext_vector_type(8)));
	constexpr u32 VPB = 256 * 8, WIN = 4096;   // vectors per workgroup and step; bins searched in the LDS
	__shared__ u64 win[WIN + 1];
	__shared__ u32 s_lo, s_hi;
	const u32 tid = threadIdx.x;
	const u64 nvec = n / 8;
	auto bin_at = [&](u64 i) {
		u32 lo = 0, hi = 65536;   // offs[lo] <= i < offs[hi]
#pragma unroll
		for (int s = 0; s < 16; ++s) {
			const u32 mid = (lo + hi) >> 1;
			if (offs[mid] <= i)
				lo = mid;
			else
				hi = mid;
		}
		return lo;
	};
	for (u64 v0 = (u64)blockIdx.x * VPB; v0 < nvec; v0 += (u64)gridDim.x * VPB) {
		const u64 v1 = v0 + VPB < nvec ? v0 + VPB : nvec;
		__syncthreads();   // (the window of the previous step has been used)
		if (tid == 0)
			s_lo = bin_at(v0 * 8);
		if (tid == 64)
			s_hi = bin_at(v1 * 8 - 1);
		__syncthreads();
		const u32 blo = s_lo, bhi = s_hi;
		const bool inlds = bhi - blo < WIN;
		if (inlds)
			for (u32 j = tid; j <= bhi - blo + 1; j += 256)
				win[j] = offs[blo + j];
		__syncthreads();
		for (u64 v = v0 + tid; v < v1; v += 256) {
			const u64 i0 = v * 8;
			u32 b;
			kvec_t x;
			if (inlds) {
				u32 lo = 0, hi = bhi - blo + 1;   // win[lo] <= i0 < win[hi]
				while (hi - lo > 1) {
					const u32 mid = (lo + hi) >> 1;
					if (win[mid] <= i0)
						lo = mid;
					else
						hi = mid;
				}
				b = lo;
				if (win[b + 1] >= i0 + 8) {
					const uint16_t k = kdf_invert<uint16_t>((uint16_t)(blo + b), ka);
#pragma unroll
					for (u32 e = 0; e < 8; ++e)
						x[e] = k;
				} else {
#pragma unroll
					for (u32 e = 0; e < 8; ++e) {
						while (win[b + 1] <= i0 + e)
							++b;
						x[e] = kdf_invert<uint16_t>((uint16_t)(blo + b), ka);
					}
				}
			} else {
				b = bin_at(i0);
#pragma unroll
				for (u32 e = 0; e < 8; ++e) {
					while (offs[b + 1] <= i0 + e)
						++b;
					x[e] = kdf_invert<uint16_t>((uint16_t)b, ka);
				}
			}
			*(kvec_t *)(out + i0) = x;
		}
	}
	if (blockIdx.x == 0 && tid < (u32)(n - nvec * 8)) {
		const u64 i = nvec * 8 + tid;
		out[i] = kdf_invert<uint16_t>((uint16_t)bin_at(i), ka);
	}
}

// dst[i] = (narrower or wider) src[i]
template <typename DT, typename ST_>
__global__ void rsx_convert_kernel(DT *dst, const ST_ *src, u64 n)
{
	const u64 stride = (u64)gridDim.x * blockDim.x;
	for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
		dst[i] = (DT)src[i];
}

// Zeroes up to three regions (16-byte granules) in one launch: the flags, the histogram and the status words of all the
// passes of a sort.  (Each hipMemsetAsync is a launch of its own; at 10^5 keys six of them were a third of the sort.)
__global__ __launch_bounds__(256) void rsx_zero3_kernel(u32x4 *a, u64 na, u32x4 *b, u64 nb, u32x4 *c, u64 nc,
                                                        const u32 *gate = nullptr)
{
	// (a device-scheduled sort that got by without the histogram: the flags hold ITS plan, which the copy home still reads)
	if (gate && *gate == GATE_DONE)
		return;
	const u64 stride = (u64)gridDim.x * blockDim.x, t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
	const u32x4 z = {0u, 0u, 0u, 0u};
	for (u64 i = t; i < na; i += stride)
		a[i] = z;
	for (u64 i = t; i < nb; i += stride)
		b[i] = z;
	for (u64 i = t; i < nc; i += stride)
		c[i] = z;
}

// dst = src if the device-side plan says that the sort ended in `src`'s buffer (an odd number of kept columns):
// the last step of rsx_sort_inplace_async.  16-byte granules + a byte tail.
__global__ __launch_bounds__(256) void rsx_copy_if_odd_kernel(unsigned char *__restrict__ dst, const unsigned char *__restrict__ src,
                                                              u64 bytes, const Plan *__restrict__ plan)
{
	if (plan->sorted || !(plan->ncols & 1))
		return;
	const u64 n16 = ((((uintptr_t)dst | (uintptr_t)src) & 15) == 0) ? bytes / 16 : 0;
	const u64 stride = (u64)gridDim.x * blockDim.x, t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
	for (u64 i = t; i < n16; i += stride)
		((u32x4 *)dst)[i] = ((const u32x4 *)src)[i];
	for (u64 i = n16 * 16 + t; i < bytes; i += stride)
		dst[i] = src[i];
}

// keys[i] = the KT at byte key_off of record i (records rec_bytes apart; any alignment: assembled from bytes unless
// both the stride and the offset are multiples of sizeof(KT))
template <typename KT>
__global__ void rsx_extract_key_kernel(KT *__restrict__ keys, const unsigned char *__restrict__ recs, u64 n, u32 rec_bytes,
                                       u32 key_off)
{
	const bool aligned = rec_bytes % sizeof(KT) == 0 && key_off % sizeof(KT) == 0 && ((uintptr_t)recs % sizeof(KT)) == 0;
	const u64 stride = (u64)gridDim.x * blockDim.x;
	for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
		const unsigned char *p = recs + i * rec_bytes + key_off;
		KT k;
		if (aligned) {
			k = *(const KT *)p;
		} else {
			k = 0;
#pragma unroll
			for (int b = 0; b < (int)sizeof(KT); ++b)
				k |= (KT)p[b] << (8 * b);
		}
		keys[i] = k;
	}
}

// dst record i = src record idx[i]; records are rec_bytes wide, moved in WORD units
template <typename WORD, typename IT>
__global__ void rsx_gather_kernel(WORD *__restrict__ dst, const WORD *__restrict__ src, const IT *__restrict__ idx,
                                  u64 n, u32 words_per_rec)
{
	const u64 total = n * words_per_rec;
	const u64 stride = (u64)gridDim.x * blockDim.x;
	for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
		const u64 rec = i / words_per_rec;
		const u32 w = (u32)(i - rec * words_per_rec);
		dst[i] = src[(u64)idx[rec] * words_per_rec + w];
	}
}

}  // namespace rsx
