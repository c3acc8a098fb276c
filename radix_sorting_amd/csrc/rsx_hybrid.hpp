// rsx_hybrid.hpp -- "one MSB pass and then LSB sort the sub-results" (README.md:647-650), gfx950.
//
// The reference's loop at radix_sort.hpp:82-90 makes one trip through memory per kept column.  The result of those P stable
// passes is the stable order by the kept columns, and that order can be reached with fewer trips when the keys spread over
// the digits of their TOP kept columns:
//
//   level 1  ONE ordinary stable scatter pass by the highest kept column (the pass kernel of rsx_scatter2.hpp, unchanged)
//            leaves 256 buckets; if every bucket fits a workgroup (leaf capacity), rsx_leaf_sort_kernel sorts each bucket by
//            the remaining kept columns, LSB first, with the keys in registers and one LDS staging area: read once, written
//            once, in place or into the other buffer -- whichever the reference's parity rule names (radix_sort.hpp:92).
//   level 2  larger arrays: a second pass by the next kept column INSIDE each bucket (the same pass kernel, SEG: its tiles are
//            cut at bucket boundaries and the look-back chain restarts in every bucket) leaves 65536 buckets for the leaves.
//            The per-bucket digit counts that pass needs come from rsx_seg_hist_kernel (one read of the pass-1 output).
//            If a (digit, digit) bucket turns out larger than a leaf -- keys clustered in their top sixteen bits -- the
//            segmented passes simply go on LSB first over the remaining columns inside each level-1 bucket (mode B): the
//            same result with one trip per column, as the reference.
//
// Stable: every pass and every leaf pass is; a bucket holds all keys with its digits in their order of arrival.  The element
// images are moved untouched (the KDF only picks digits), so the output is bit-identical to the P-pass sort.
//
// Which way a sort goes is decided on the device (rsx_plan_kernel -> Plan::hyb, from the column histograms it has anyway;
// rsx_seg_plan_kernel -> SegCtl::mode) and read by the host where it has to enqueue different kernels.
#pragma once

#include "rsx_kernels.hpp"

namespace rsx {

enum : u32 { HYB_NONE = 0, HYB_ONE_LEVEL = 1, HYB_TWO_LEVEL = 2 };
enum : u32 { SEG_MODE_NONE = 0, SEG_MODE_LEAVES = 1, SEG_MODE_LSD = 2, SEG_MODE_RETRY = 3 };
static_assert(GATE_DONE == SEG_MODE_LEAVES, "the gate of the histogram-first kernels is SegCtl::mode");

// One tile of a segmented pass: [beg, beg + cnt) lies inside level-1 bucket `bucket`; `first` is the index of the bucket's
// first tile (where the look-back chain of the bucket ends).
struct SegTile {
	u32 beg, cnt, bucket, first;
};

// Device-side control block of the level-2 part of a sort (zeroed by the host before rsx_seg_tiles_kernel).
struct SegCtl {
	u32 ntiles;     // rsx_seg_tiles_kernel
	u32 mode;       // rsx_seg_plan_kernel: SEG_MODE_LEAVES (all (digit, digit) buckets fit a leaf) or SEG_MODE_LSD
	u32 maxleaf;    // the largest (digit, digit) bucket
	u32 done;       // blocks of rsx_seg_plan_kernel that are through
	u32 nleaf;      // leaves in segtab (rsx_seg_plan_kernel)
	u32 overflow;   // slack attempt: a slot was too small (rsx_scatter2_kernel, SCATTER_SEG_SLACK)
	u32 blind;      // sorts without a histogram (rsx_blind_precheck_kernel): BLIND_GO while nothing speaks against going on
	// ... the byte columns the sample found constant (0xFF per column, derived-key space) and the first key's derived key: the
	// level-1 pass checks EVERY key against them (a column is skipped only if all keys share its byte, radix_sort.hpp:64-70)
	u32 cmask_lo, cmask_hi, key0_lo, key0_hi;
	u32 nredo;      // leaves rsx_leaf16_kernel (rsx_leaf16.hpp) left to rsx_leaf_sort_kernel: entries of its `redo` list
	u32 leaf16;     // rsx_blind_precheck_kernel: the sampled keys spread over the top twelve of their low sixteen bits (rsx_leaf16_kernel's bins)
	// sorts without a histogram: the bit positions of the two MSB passes' 8-bit digits.  8 x the two highest kept columns, as the
	// reference's bytes (radix_sort.hpp:40-45) -- or, for 4-byte keys whose top bits are the same in every key (shift1 + 8 < 32:
	// values below 2^30, one rank's share of a distributed sort), the sixteen bits below the highest bit that varies: the
	// order is the same (the bits above are constant, checked on every key through cmask) and the buckets are even again
	u32 shift1, shift2;
	// 8-byte keys whose leaves sort columns of the low word only (keys below 2^40, say): the level-2 pass writes the low word
	// of every DERIVED key into its slots (four bytes per key instead of eight) and rsx_leafk_kernel's SLOT32 form reads them;
	// the upper word comes back from key0, the constant columns and the slot's two digits.  2: the level-1 slots hold low words
	// too (nothing below the level-1 digit varies above bit 32: rsx_pass32a_kernel's OT = u32 form, rsx_pass64a_kernel<u32, u32>)
	u32 narrow;
	// Rank sorts of 4-byte keys whose VARYING bits are few (README.md:716-758, key compaction: f32 & 0xFFF000FF, BASELINE.json's
	// cfg 4 (iii), varies in 20 bits -- and its byte columns hold 2, 32 and 256 values, which no slot scheme by bytes takes):
	// the level-1 pass packs the bits in which the sampled keys differ from the first key into the low `compact` bits of an
	// unsigned key (up to four runs of bits: cpiece[] = source shift | width << 8 | destination shift << 16; the KDF's flips applied
	// in the packed space: dropping bits that are the same in every key does not change the order), writes THAT into its slots, and
	// every kernel behind it sorts plain unsigned keys whose digits lie below bit `compact` (shift1 / shift2).  A key that
	// differs from the first one (craw0, as the caller wrote it) outside the varying bits (cvnot) calls the attempt off, as a
	// column taken for constant that is not (cmask) does for the byte scheme.  0: no compaction.
	u32 compact;
	u32 cpiece[4];
	u32 craw0, cvnot;
	// ckind == 1: FLOATS ON A GRID.  f32 keys that are all whole multiples of one power of two and smaller than another --
	// measurements, prices, normalised values: BASELINE.json's cfg 4 (ii), (int24 - 2^23) x 2^-23, whose sign-and-exponent byte
	// holds 87 % of the keys in four values -- are fixed-point numbers: key x 2^-e0 (cpiece[0]: that factor's bits) is an integer
	// of `compact` bits, in the keys' order, and spreads as evenly as the values do.  The level-1 pass converts (and checks every
	// key: finite, on the grid, in range, not -0.0, which the reference orders before +0.0 and an integer cannot).
	u32 ckind;
	// Device-scheduled sorts (rsx_sort_inplace_async): nobody reads a verdict back, so the back-off the host keeps for the blocking
	// sorts lives here.  An attempt that is LOST after its sample let it through -- a slot overflowed, a key differed in a column
	// taken for constant: one or two full passes wasted -- sets boff_skip = boff_next = 1, 2, 4 .. 64: the next boff_skip sorts in
	// this context do not try (the sample kernel says no at once); an attempt that goes through resets boff_next.  Never zeroed
	// by the sample kernel; zeroed where the control block is allocated.
	u32 boff_skip, boff_next;
};
__device__ __forceinline__ void segctl_attempt_lost(SegCtl *ctl)
{
	const u32 nx = ctl->boff_next ? (ctl->boff_next < 32u ? 2u * ctl->boff_next : 64u) : 1u;
	ctl->boff_next = nx;
	ctl->boff_skip = nx;
}
enum : u32 { BLIND_NONE = 0, BLIND_GO = 1, BLIND_FAILED = 2 };

// the packed key of `raw` (SegCtl::compact): the varying bits' runs moved together, then the KDF's flips in the packed space
// (fneg: a float key with its sign bit set; top: the packed position of the sign bit, if it varies; desc: descending order)
template <typename KT>
__device__ __forceinline__ u32 compact_key(u32 raw, const u32 (&piece)[4], u32 nb, const KdfArgs<KT> ka)
{
	u32 c = 0;
#pragma unroll
	for (int i = 0; i < 4; ++i) {
		const u32 p = piece[i];
		c |= __builtin_amdgcn_ubfe(raw, p & 31u, (p >> 8) & 63u) << (p >> 16);
	}
	const u32 all = nb >= 32u ? ~0u : (1u << nb) - 1u;
	// which packed bits the KDF flips: every one for a negative float and for descending order; the sign bit's packed place
	// (the top packed bit, if the sign varies at all) for signed keys and non-negative floats
	const bool neg = ka.fmask != 0 && (raw >> 31) != 0;
	const u32 top = (piece[0] & 31u) + ((piece[0] >> 8) & 63u) == 32u ? 1u << (nb - 1u) : 0u;   // (piece 0 is the highest run)
	u32 flip = neg ? all : (ka.sflip ? top : 0u);
	if (ka.desc)
		flip ^= all;
	return c ^ flip;
}

// A leaf's keys: [beg, beg + cnt), sorted by the `ncols` lowest kept columns.  Level 2: a (digit, digit) bucket (ncols = all
// columns below the level-2 one) or a run of small neighbouring ones of the same level-1 bucket (one column more).
struct LeafSeg {
	u32 beg, cnt, ncols;
	u32 slot;   // 0: the keys lie at `beg` of the input buffer; s + 1: in slot s of the scratch array (slack attempt)
};

constexpr u32 LEAF_MERGE_CAP = 4096;   // neighbouring (digit, digit) buckets are sorted together while they hold no more keys than this

// exclusive scan of one u32 per thread over a workgroup of 256 threads (4 waves); `tot` = the sum
__device__ __forceinline__ u32 block_scan_256(u32 v, u32 *s_w, u32 &tot)
{
	const u32 lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
	u32 x = v;
#pragma unroll
	for (int off = 1; off < 64; off <<= 1) {
		const u32 y = __shfl_up(x, off);
		if (lane >= (u32)off)
			x += y;
	}
	if (lane == 63)
		s_w[wid] = x;
	__syncthreads();
	u32 base = 0;
#pragma unroll
	for (int k = 0; k < 4; ++k) {
		if (k < (int)wid)
			base += s_w[k];
	}
	tot = s_w[0] + s_w[1] + s_w[2] + s_w[3];
	__syncthreads();
	return base + x - v;
}

// the fixed-point key of the float `raw` (SegCtl::ckind == 1): raw x scale as an integer of nb bits, offset to unsigned; `bad` is set
// for a key that is not finite, not on the grid, out of range, or -0.0
// (nonneg: no key is negative -- the sample saw none, every key is checked --: the integers are the keys, no offset, one bit less)
__device__ __forceinline__ u32 fixedpoint_key(u32 raw, u32 scale_bits, u32 nb, bool nonneg, bool desc, u32 &bad)
{
	const float tf = __uint_as_float(raw) * __uint_as_float(scale_bits);   // (exact: a power of two; overflow -> inf fails the range test)
	const float hi = __uint_as_float((127u + nb - (nonneg ? 0u : 1u)) << 23);   // 2^nb or 2^(nb - 1)
	const float lo = nonneg ? 0.0f : -hi;
	const int t = (int)tf;
	if (!(tf >= lo && tf < hi) || (float)t != tf || raw == 0x80000000u)   // (NaN fails the first test)
		bad |= 1u;
	u32 key = nonneg ? (u32)t : (u32)(t + (int)(1u << (nb - 1u)));
	if (desc)
		key = ~key & ((1u << nb) - 1u);
	return key;
}

// ---- tiles of the segmented passes ----------------------------------------------------------------------------------
// off1 = the exclusive offsets of the level-1 column (ghist + 256 * c1, after rsx_plan_kernel).  Every workgroup scans the 256
// bucket sizes (cheap) and writes its share of the tiles.
__global__ __launch_bounds__(256) void rsx_seg_tiles_kernel(const u64 *__restrict__ ghist, u64 n, const Plan *__restrict__ plan,
                                                            u32 tile, SegTile *__restrict__ tiles, SegCtl *__restrict__ ctl,
                                                            u32 *__restrict__ btile,   // [257]: bucket k's tiles are [btile[k], btile[k + 1])
                                                            u64 *__restrict__ off1_out = nullptr, u32 blind_cap = 0,
                                                            const u32 *__restrict__ status0 = nullptr, u32 ntiles0 = 0,
                                                            u32 back_cap = 0, u32 tile_narrow = 0,   // (tile_narrow: the tile of the level-2 pass that runs when SegCtl::narrow is set, rsx_pass64.hpp)
                                                            u32 tile_narrow2 = 0)                    // (... and when it is 2: the form that reads four-byte values)
{
	// blind_cap != 0 (a sort without a histogram): no offsets exist.  The inclusive prefix of the LAST tile of the level-1 pass
	// (rsx_scatter2_kernel, SCATTER_BLIND_TOP; status0) is the size of every bucket; bucket k lies in ITS SLOT of blind_cap keys
	// of that pass's output.  The sizes' exclusive scan -- the buckets' places in the dense result, what `ghist` would hold
	// (radix_sort.hpp:72-80) -- goes to off1_out[256]; a slot that overflowed ends the attempt.
	if (plan->hyb != HYB_TWO_LEVEL || (blind_cap && ctl->blind != BLIND_GO))
		return;
	if (tile_narrow && ctl->narrow)
		tile = (tile_narrow2 && ctl->narrow == 2u) ? tile_narrow2 : tile_narrow;
	__shared__ u32 s_size[256], s_tb[257], s_beg[256], s_w[4], s_back[256];
	const u32 d = threadIdx.x;
	u32 size, back = 0;
	if (blind_cap) {
		if (back_cap) {
			// back_cap != 0: the level-1 pass was rsx_pass32a_kernel (rsx_pass32.hpp) -- no chain; bucket d lies at BOTH ends of its slot:
			// status0[d] keys from the slot's beginning on, status0[256 + d] in its last back_cap places (the slot's two cursors)
			size = __hip_atomic_load(status0 + d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			back = __hip_atomic_load(status0 + 256 + d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			size += back;
		} else {
			size = __hip_atomic_load(status0 + ((u64)(ntiles0 - 1) * 256 + d), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & StatusBits<u32>::VALMASK;
		}
		const bool over = size > blind_cap || ctl->overflow != 0;
		u32 tot1;
		const u32 o = block_scan_256(size, s_w, tot1);
		if (blockIdx.x == 0)
			off1_out[d] = o;
		if (__syncthreads_or(over ? 1 : 0)) {
			if (blockIdx.x == 0 && d == 0) {
				ctl->blind = BLIND_FAILED;
				segctl_attempt_lost(ctl);
			}
			return;
		}
		s_beg[d] = d * blind_cap;
	} else {
		const u64 *off1 = ghist + 256 * plan->cols[plan->ncols - 1];
		const u64 b = off1[d], e = d == 255 ? n : off1[d + 1];
		size = (u32)(e - b);
		s_beg[d] = (u32)b;
	}
	s_size[d] = size;
	s_back[d] = back;
	u32 total;
	// (a bucket's tiles: those of its front part, then ONE for what lies at the slot's end -- at most back_cap keys, less than a tile)
	const u32 tb = block_scan_256((size - back + tile - 1) / tile + (back ? 1u : 0u), s_w, total);
	s_tb[d] = tb;
	if (d == 0) {
		s_tb[256] = total;
		if (blockIdx.x == 0) {
			ctl->ntiles = total;
			btile[256] = total;
		}
	}
	if (blockIdx.x == 0)
		btile[d] = tb;
	__syncthreads();
	for (u32 t = blockIdx.x * 256 + d; t < total; t += gridDim.x * 256) {
		u32 lo = 0, hi = 256;   // the bucket k with s_tb[k] <= t < s_tb[k + 1] (empty buckets have s_tb[k] == s_tb[k + 1])
		while (hi - lo > 1) {
			const u32 mid = (lo + hi) >> 1;
			if (s_tb[mid] <= t)
				lo = mid;
			else
				hi = mid;
		}
		while (s_tb[lo + 1] <= t)   // (skip empty buckets that share the boundary)
			++lo;
		const u32 k = lo, j = t - s_tb[k];
		SegTile st;
		const u32 front = s_size[k] - s_back[k];
		if (j * tile < front) {
			st.beg = s_beg[k] + j * tile;
			const u32 left = front - j * tile;
			st.cnt = left < tile ? left : tile;
		} else {
			st.beg = s_beg[k] + blind_cap - back_cap;
			st.cnt = s_back[k];
		}
		st.bucket = k;
		st.first = s_tb[k];
		tiles[t] = st;
	}
}

// ---- digit counts per level-1 bucket ------------------------------------------------------------------------------------
// seghist[bucket][slot][256], slot k = the k-th kept column (all kept columns but the level-1 one).  A workgroup takes a
// contiguous range of tiles (tiles never straddle a bucket), counts into LDS and adds its counts to the bucket's rows when the
// bucket changes.
// This kernel: the slots below the level-2 column, wanted only in SEG_MODE_LSD (launched once the host knows);
// rsx_seg_hist1_kernel below: the level-2 column alone, which every two-level sort needs.
template <typename KT>
__global__ __launch_bounds__(1024) void rsx_seg_hist_kernel(const KT *__restrict__ keys, const SegTile *__restrict__ tiles,
                                                            const SegCtl *__restrict__ ctl, const Plan *__restrict__ plan,
                                                            KdfArgs<KT> ka, u32 *__restrict__ seghist)
{
	if (plan->hyb != HYB_TWO_LEVEL || ctl->mode != SEG_MODE_LSD)
		return;
	constexpr int MAXS = sizeof(KT) - 1;
	__shared__ u32 h[MAXS][256];
	const u32 tid = threadIdx.x;
	const u32 nslots = plan->ncols - 2;   // (the level-2 column's row is there already)
	u32 shifts[MAXS];
#pragma unroll
	for (int k = 0; k < MAXS; ++k)
		shifts[k] = 8 * plan->cols[k < (int)nslots ? k : 0];
	const u32 nt = ctl->ntiles;
	const u32 t0 = (u32)((u64)nt * blockIdx.x / gridDim.x), t1 = (u32)((u64)nt * (blockIdx.x + 1) / gridDim.x);
	if (t0 == t1)
		return;
	for (u32 i = tid; i < MAXS * 256; i += 1024)
		(&h[0][0])[i] = 0;
	__syncthreads();
	u32 bucket = tiles[t0].bucket;
	auto flush = [&]() {
		__syncthreads();
		for (u32 i = tid; i < nslots * 256; i += 1024) {
			const u32 v = (&h[0][0])[i];
			if (v)
				atomicAdd(&seghist[(u64)bucket * (MAXS * 256) + i], v);
			(&h[0][0])[i] = 0;
		}
		__syncthreads();
	};
	for (u32 t = t0; t < t1; ++t) {
		const SegTile st = tiles[t];
		if (st.bucket != bucket) {
			flush();
			bucket = st.bucket;
		}
		const KT *p = keys + st.beg;
		constexpr int U = 8;
		for (u32 i0 = 0; i0 < st.cnt; i0 += 1024 * U) {
			KT v[U];
#pragma unroll
			for (int u = 0; u < U; ++u) {
				const u32 i = i0 + u * 1024 + tid;
				v[u] = i < st.cnt ? p[i] : (KT)0;
			}
#pragma unroll
			for (int u = 0; u < U; ++u) {
				const u32 i = i0 + u * 1024 + tid;
				if (i < st.cnt) {
					const KT k = kdf_apply(v[u], ka);
#pragma unroll
					for (int s = 0; s < MAXS; ++s)
						if (s < (int)nslots)
							atomicAdd(&h[s][(u32)(k >> shifts[s]) & 0xFFu], 1u);
				}
			}
		}
	}
	flush();
}

// The level-2 column's digit counts per level-1 bucket: one LDS atomic per key on the wave's own row of 256 counters.
template <typename KT>
__global__ __launch_bounds__(1024) void rsx_seg_hist1_kernel(const KT *__restrict__ keys, const SegTile *__restrict__ tiles,
                                                             const SegCtl *__restrict__ ctl, const Plan *__restrict__ plan,
                                                             KdfArgs<KT> ka, u32 *__restrict__ seghist)
{
	if (plan->hyb != HYB_TWO_LEVEL)
		return;
	constexpr int MAXS = sizeof(KT) - 1, NW = 16;
	__shared__ u32 h[NW][256];
	const u32 tid = threadIdx.x, wid = tid >> 6;
	const u32 slot = plan->ncols - 2;
	const u32 shift = 8 * plan->cols[slot];
	const u32 nt = ctl->ntiles;
	const u32 t0 = (u32)((u64)nt * blockIdx.x / gridDim.x), t1 = (u32)((u64)nt * (blockIdx.x + 1) / gridDim.x);
	if (t0 == t1)
		return;
	for (u32 i = tid; i < NW * 256; i += 1024)
		(&h[0][0])[i] = 0;
	__syncthreads();
	u32 bucket = tiles[t0].bucket;
	auto flush = [&]() {
		__syncthreads();
		if (tid < 256) {
			u32 v = 0;
#pragma unroll
			for (int w = 0; w < NW; ++w) {
				v += h[w][tid];
				h[w][tid] = 0;
			}
			if (v)
				atomicAdd(&seghist[((u64)bucket * MAXS + slot) * 256 + tid], v);
		}
		__syncthreads();
	};
	u32 *hw = h[wid];
	for (u32 t = t0; t < t1; ++t) {
		const SegTile st = tiles[t];
		if (st.bucket != bucket) {
			flush();
			bucket = st.bucket;
		}
		const KT *p = keys + st.beg;
		constexpr int U = 8;
		for (u32 i0 = 0; i0 < st.cnt; i0 += 1024 * U) {
			KT v[U];
			if (i0 + 1024 * U <= st.cnt) {
#pragma unroll
				for (int u = 0; u < U; ++u)
					v[u] = p[i0 + u * 1024 + tid];
#pragma unroll
				for (int u = 0; u < U; ++u)
					atomicAdd(&hw[digit_of(v[u], ka, shift)], 1u);
			} else {
#pragma unroll
				for (int u = 0; u < U; ++u) {
					const u32 i = i0 + u * 1024 + tid;
					v[u] = i < st.cnt ? p[i] : (KT)0;
				}
#pragma unroll
				for (int u = 0; u < U; ++u) {
					const u32 i = i0 + u * 1024 + tid;
					if (i < st.cnt)
						atomicAdd(&hw[digit_of(v[u], ka, shift)], 1u);
				}
			}
		}
	}
	flush();
}

// ---- per-bucket exclusive scans, the leaves' segments, and the decision -----------------------------------------------------
// One workgroup per level-1 bucket, thread = digit.  seghist rows become exclusive offsets relative to the bucket's start
// (radix_sort.hpp:72-80 per bucket); segtab[bucket * 256 + digit] = the (digit, digit) bucket of the level-2 column.
template <typename KT>
__global__ __launch_bounds__(256) void rsx_seg_plan_kernel(u32 *__restrict__ seghist, const u64 *__restrict__ ghist, u64 n,
                                                           const Plan *__restrict__ plan, SegCtl *__restrict__ ctl,
                                                           LeafSeg *__restrict__ segtab, u32 leaf_cap, SegCtl *host_ctl,
                                                           u32 phase)
{
	// phase 0: the level-2 column's row, the leaves and the decision; phase 1 (SEG_MODE_LSD only, enqueued once the host
	// knows): the rows of the columns below it (rsx_seg_hist_kernel has counted them by then)
	if (plan->hyb != HYB_TWO_LEVEL || (phase == 1 && ctl->mode != SEG_MODE_LSD))
		return;
	constexpr int MAXS = sizeof(KT) - 1;
	__shared__ u64 tot[256];
	__shared__ u64 lsum[64];
	__shared__ u32 s_max;
	const u32 d = threadIdx.x, b = blockIdx.x;
	const u32 nslots = plan->ncols - 1;
	const u64 *off1 = ghist + 256 * plan->cols[plan->ncols - 1];
	const u64 bbeg = off1[b];
	if (d == 0)
		s_max = 0;
	for (u32 s = phase ? 0 : nslots - 1; s < (phase ? nslots - 1 : nslots); ++s) {
		u32 *row = seghist + ((u64)b * MAXS + s) * 256;
		const u32 c = row[d];
		__syncthreads();
		tot[d] = c;
		__syncthreads();
		if (d < 64)
			wave_scan_256(tot, lsum, d);
		__syncthreads();
		row[d] = (u32)tot[d];
		if (s == nslots - 1) {   // the level-2 column: the highest of the remaining ones
			// The leaves of this bucket: every (digit, digit) bucket, small neighbours taken together (a workgroup per
			// 100-key bucket is all overhead).  A bucket opens a new leaf unless it and its predecessor are small (at most
			// half the merge cap) and start in the same window of half the cap: a merged leaf then stays below the cap.
			__shared__ u32 s_cnt[256], s_start[257], s_first[257], s_w[4];
			constexpr u32 H = LEAF_MERGE_CAP / 2;
			const u32 o = (u32)tot[d];
			s_cnt[d] = c;
			__syncthreads();
			const bool opens = d == 0 || c > H || s_cnt[d - 1] > H || (o / H) != ((o - s_cnt[d - 1]) / H);
			u32 nleaf;
			const u32 idx = block_scan_256(opens ? 1u : 0u, s_w, nleaf);
			if (opens) {
				s_start[idx] = o;
				s_first[idx] = d;
			}
			if (d == 0) {
				s_start[nleaf] = o;   // (overwritten below by the thread that knows the total)
				s_first[nleaf] = 256;
			}
			__syncthreads();
			if (d == 255)
				s_start[nleaf] = o + c;
			__syncthreads();
			// (the table is dense: a bucket's leaves take the next free slots; their order across buckets does not matter)
			__shared__ u32 s_slot;
			if (d == 0)
				s_slot = atomicAdd(&ctl->nleaf, nleaf);
			__syncthreads();
			u32 mine = 0;
			if (d < nleaf) {
				LeafSeg ls;
				ls.beg = (u32)(bbeg + s_start[d]);
				ls.cnt = s_start[d + 1] - s_start[d];
				ls.ncols = nslots - 1 + (s_first[d + 1] - s_first[d] > 1 ? 1u : 0u);
				ls.slot = 0;
				segtab[s_slot + d] = ls;
				mine = ls.cnt;
			}
			atomicMax(&s_max, mine);
		}
	}
	if (phase)
		return;
	__syncthreads();
	__shared__ u32 s_last;
	if (d == 0) {
		atomicMax(&ctl->maxleaf, s_max);
		__threadfence();
		s_last = atomicAdd(&ctl->done, 1u) == gridDim.x - 1 ? 1u : 0u;
	}
	__syncthreads();
	if (s_last && d == 0) {
		__threadfence();
		const u32 mx = atomicMax(&ctl->maxleaf, 0u);
		const u32 mode = mx <= leaf_cap ? SEG_MODE_LEAVES : SEG_MODE_LSD;
		ctl->mode = mode;
		if (host_ctl) {
			host_ctl->ntiles = ctl->ntiles;
			host_ctl->maxleaf = mx;
			host_ctl->nleaf = ctl->nleaf;
			host_ctl->mode = mode;
			__threadfence_system();
		}
	}
}

// ---- slack attempt: the (digit, digit) counts read off the finished chain ---------------------------------------------
// After a SCATTER_SEG_SLACK pass the inclusive prefix of a bucket's LAST tile is the bucket's count of every level-2 digit.
// One workgroup per level-1 bucket: the leaves (one per (digit, digit) bucket, read from its slot, written to its place in the
// dense output; table entry b * 256 + d, empty buckets stay in the table with no keys) and the verdict: SEG_MODE_LEAVES if no
// slot overflowed, else SEG_MODE_RETRY (the host runs the counted path from the untouched pass-1 output).  The pass itself
// flags every run that leaves its slot, so the verdict is there when this kernel starts and no workgroup waits for another
// (with a table compacted through a global counter and a last-workgroup-decides counter this kernel took 14.6 us: three
// serialised same-address atomics per workgroup).
template <typename ST>
__global__ __launch_bounds__(256) void rsx_seg_slack_plan_kernel(const ST *__restrict__ status, const u32 *__restrict__ btile,
                                                                 const u64 *__restrict__ ghist, const Plan *__restrict__ plan,
                                                                 SegCtl *__restrict__ ctl, LeafSeg *__restrict__ segtab,
                                                                 u32 slack_cap, SegCtl *host_ctl,
                                                                 const u64 *__restrict__ off1_given = nullptr, u32 blind_ = 0)
{
	u32 blind = blind_ & 0xFFu;
	const u32 rev = blind_ >> 8;   // (rev: the table in reverse slot order -- the leaves then start with the slots written last)
	if (blind == 3u)               // (8-byte keys: the pass into four-byte slots keeps cursors, rsx_pass64.hpp, the one into whole-key slots a chain)
		blind = ctl->narrow ? 2u : 1u;
	if (plan->hyb != HYB_TWO_LEVEL || (blind && ctl->blind != BLIND_GO))
		return;
	typedef StatusBits<ST> SB_;
	__shared__ u32 s_w[4];
	const u32 d = threadIdx.x, b = blockIdx.x;
	const u64 *off1 = off1_given ? off1_given : ghist + 256 * plan->cols[plan->ncols - 1];
	const u32 t0 = btile[b], t1 = btile[b + 1];
	const u32 ncols = plan->ncols;
	const u64 bbeg = off1[b];
	u32 c = 0, back = 0;
	if (blind == 2) {
		// rsx_pass16a_kernel (rsx_pass16.hpp): no chain -- the slot's two cursors, its front's and its back's, at the beginning of
		// the status region
		c = (u32)__hip_atomic_load(status + ((u64)b * 256 + d), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		back = (u32)__hip_atomic_load(status + ((u64)65536 + b * 256 + d), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		c += back;
	} else if (t1 > t0) {
		c = (u32)(__hip_atomic_load(status + ((u64)(t1 - 1) * 256 + d), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & SB_::VALMASK);
	}
	if (b == 0 && d == 0) {
		const u32 mode = ctl->overflow == 0 ? SEG_MODE_LEAVES : SEG_MODE_RETRY;
		ctl->mode = mode;
		ctl->nleaf = 65536;
		ctl->maxleaf = slack_cap;   // (no leaf is larger: a run that leaves its slot sets the flag)
		if (blind) {
			if (mode == SEG_MODE_LEAVES)
				ctl->boff_next = 0;
			else
				segctl_attempt_lost(ctl);
		}
		if (host_ctl) {
			host_ctl->ntiles = ctl->ntiles;
			host_ctl->maxleaf = slack_cap;
			host_ctl->nleaf = 65536;
			host_ctl->overflow = ctl->overflow;
			host_ctl->narrow = ctl->narrow;
			host_ctl->mode = mode;
		}
	}
	u32 total;
	const u32 o = block_scan_256(c, s_w, total);
	LeafSeg ls;
	ls.beg = (u32)(bbeg + o);
	ls.cnt = c <= slack_cap ? c : 0u;   // (an overflowed slot: the attempt is discarded anyway)
	ls.ncols = (ncols - 2) | (back << 16);   // (the upper half: how many of the values lie at the slot's end, rsx_leaf16_kernel)
	ls.slot = b * 256 + d + 1;
	segtab[rev ? 65535u - (b * 256 + d) : b * 256 + d] = ls;
}

// ---- sorts without a histogram ("blind") ------------------------------------------------------------------------------------
// The histogram of radix_sort.hpp:47-58 serves three purposes: the pre-sorted exit (:60-62), the kept columns (:64-70) and the
// offsets (:72-80).  For keys that spread evenly over ALL their columns -- BASELINE.json's headline, 2^28 uniform u32 keys --
// a two-level sort needs none of its counts: both MSB passes write into slots of 1.25 times the expected bucket size and the
// bucket sizes are read off the look-back chains.  What is left of the histogram's job is decided EXACTLY from a sample:
//   * one descent among sampled neighbours proves the input unsorted (no early exit to honour);
//   * two different bytes among the samples of a column prove the column kept; a column whose samples all agree is taken
//     for constant, and the level-1 pass -- which sees every key anyway -- compares that byte of every key with the first
//     key's (SegCtl::cmask / key0): one key that differs calls the attempt off like an overflowing slot does.  So the list
//     of kept columns (radix_sort.hpp:64-70), and with it the buffer the result lies in (:92), is exact.
// If the sample proves both, and shows no sign of clustering (which would only cost the attempt: a slot that overflows sets
// SegCtl::overflow and everything after it is skipped; the caller's array is only READ until the leaves, so the ordinary
// histogram-first sort then starts from untouched input), the sort goes on without the 0.24 ms (of 1.85) the histogram's
// read of 2^28 keys costs.  Otherwise the host runs the ordinary path and remembers not to try for a while.
// One workgroup; 64 places spread over the array, 128 consecutive keys (8 per thread) at each: every place is an address
// translation of its own, and 1024 scattered places took 14.6 us where this takes a third of that.
// Workgroups 1 .. : zero the status words of the two passes (z, nz 16-byte words) -- one launch for both jobs.
template <typename KT>
__global__ __launch_bounds__(1024) void rsx_blind_precheck_kernel(const KT *__restrict__ src, u64 n, KdfArgs<KT> ka,
                                                                  SegCtl *__restrict__ ctl, Plan *__restrict__ plan,
                                                                  Plan *host_plan, u32x4 *__restrict__ z, u64 nz, u32 min_cols,
                                                                  u32 allow_shift = 0, u32 allow_narrow = 0, u32 allow_compact = 0,
                                                                  u32 backoff = 0,   // (1: a device-scheduled sort, SegCtl::boff_skip)
                                                                  u32 hints = 0)     // (1: the caller has COUNTED the level-1 digits and found them even)
{
	constexpr u32 W = sizeof(KT), S = 8, NS = 1024 * S;
	__shared__ u32 h[W][256];
	__shared__ u32 s_desc, s_distinct[W], s_max[W];
	// the sample's counts over the bins of the keys-only leaves that place by twelve bits (rsx_leaf16.hpp: the leaf's highest
	// column and the top nibble of the one below it -- for 4-byte keys with four kept columns the top twelve of the low sixteen bits)
	__shared__ u32 h12[4096];
	__shared__ u32 s_max12;
	__shared__ u32 hs[2][256];         // the sample's counts over the two MSB digits at their bit positions (SegCtl::shift1 / shift2)
	__shared__ u32 s_vary, s_maxs[2];  // the bits in which sampled keys differ from the first key
	const u32 tid = threadIdx.x;
	if (blockIdx.x != 0) {
		const u32x4 zero = {0, 0, 0, 0};
		for (u64 i = (u64)(blockIdx.x - 1) * 1024 + tid; i < nz; i += (u64)(gridDim.x - 1) * 1024)
			z[i] = zero;
		return;
	}
	for (u32 i = tid; i < W * 256; i += 1024)
		(&h[0][0])[i] = 0;
	if (tid < W) {
		s_distinct[tid] = 0;
		s_max[tid] = 0;
	}
	if (tid == 0)
		s_desc = s_max12 = s_vary = s_maxs[0] = s_maxs[1] = 0;
#pragma unroll
	for (u32 i = 0; i < 4; ++i)
		h12[tid + 1024 * i] = 0;
	if (tid < 512)
		(&hs[0][0])[tid] = 0;
	__syncthreads();
	const u64 i0 = ((n - 16 * S) / 63) * (tid >> 4) + (tid & 15u) * S;   // (n >= 2^20: the places do not overlap)
	KT k[S];
#pragma unroll
	for (u32 e = 0; e < S; ++e)
		k[e] = kdf_apply(src[i0 + e], ka);
	bool desc = false;
#pragma unroll
	for (u32 e = 0; e < S; ++e) {
		if (e)
			desc |= k[e] < k[e - 1];
#pragma unroll
		for (u32 c = 0; c < W; ++c)
			atomicAdd(&h[c][(u32)(k[e] >> (8 * c)) & 0xFFu], 1u);
	}
	if (__ballot(desc) && (tid & 63) == 0)
		s_desc = 1;
	if constexpr (W == 4) {
		// (all samples against the wave's first: what differs from the array's first key differs from that one or it does)
		const u32 kw = (u32)__builtin_amdgcn_readfirstlane((int)(u32)k[0]);
		u32 v = 0;
#pragma unroll
		for (u32 e = 0; e < S; ++e)
			v |= (u32)k[e] ^ kw;
#pragma unroll
		for (int off = 32; off > 0; off >>= 1)
			v |= (u32)__shfl_xor((int)v, off);
		if ((tid & 63) == 0)
			atomicOr(&s_vary, v | (kw ^ (u32)kdf_apply(src[0], ka)));
	}
	__syncthreads();
	if (tid < 256) {
#pragma unroll
		for (u32 c = 0; c < W; ++c) {
			const u32 v = h[c][tid];
			const u64 m = __ballot(v != 0);
			u32 mx = v;   // (the wave's maximum first: 64 lanes on one LDS word take their turns one by one)
#pragma unroll
			for (int off = 32; off > 0; off >>= 1) {
				const u32 y = __shfl_xor(mx, off);
				mx = y > mx ? y : mx;
			}
			if ((tid & 63) == 0) {
				atomicAdd(&s_distinct[c], (u32)__popcll(m));
				atomicMax(&s_max[c], mx);
			}
		}
	}
	__syncthreads();
	u32 shift1 = 0, shift2 = 0;
	{
		// every thread: the columns the sample proved kept, the MSB digits' bit positions, then its samples' digits and bins
		u32 nk = 0, ck[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
		for (u32 c = 0; c < W; ++c) {
			if (s_distinct[c] >= 2)
				ck[nk++] = c;
		}
		bool shifted = false;
		if (nk >= 2) {
			shift1 = 8 * ck[nk - 1];
			shift2 = 8 * ck[nk - 2];
		}
		if constexpr (W == 4) {
			// 4-byte keys, all four columns kept, the top bits constant in the sample: digits below the highest varying bit
			// (at least eight bits stay for the leaves)
			const u32 hb = s_vary ? 31u - (u32)__builtin_clz(s_vary) : 0u;
			if (allow_shift && nk == 4 && hb < 31u && hb >= 23u) {
				shift1 = hb - 7;
				shift2 = hb - 15;
				shifted = true;
			}
		}
		if (nk >= 4) {
#pragma unroll
			for (u32 e = 0; e < S; ++e) {
				atomicAdd(&hs[0][(u32)(k[e] >> shift1) & 0xFFu], 1u);
				atomicAdd(&hs[1][(u32)(k[e] >> shift2) & 0xFFu], 1u);
			}
			if (shifted) {
				// the leaves' bins: the top twelve of the shift2 bits below the MSB digits
				const u32 lowmask = (1u << shift2) - 1u, d = shift2 > 12u ? shift2 - 12u : 0u;
#pragma unroll
				for (u32 e = 0; e < S; ++e)
					atomicAdd(&h12[((u32)k[e] & lowmask) >> d], 1u);
			} else {
				const u32 sh_hi = 8 * ck[nk - 3], sh_nx = 8 * ck[nk - 4] + 4;
#pragma unroll
				for (u32 e = 0; e < S; ++e)
					atomicAdd(&h12[(((u32)(k[e] >> sh_hi) & 0xFFu) << 4) | ((u32)(k[e] >> sh_nx) & 0xFu)], 1u);
			}
		}
		__syncthreads();
		u32 m = 0;
#pragma unroll
		for (u32 i = 0; i < 4; ++i)
			m = m > h12[tid + 1024 * i] ? m : h12[tid + 1024 * i];
		u32 ms = tid < 512 ? (&hs[0][0])[tid] : 0u;   // (a wave lies inside one of the two digit tables)
#pragma unroll
		for (int off = 32; off > 0; off >>= 1) {
			const u32 y = __shfl_xor(m, off), ys = __shfl_xor(ms, off);
			m = y > m ? y : m;
			ms = ys > ms ? ys : ms;
		}
		if ((tid & 63) == 0) {
			atomicMax(&s_max12, m);
			if (tid < 512)
				atomicMax(&s_maxs[tid >> 8], ms);
		}
		__syncthreads();
	}
	// ---- rank sorts of 4-byte keys (allow_compact): would the keys' VARYING bits, packed together, spread evenly?  (SegCtl::compact)
	__shared__ u32 s_vraw, s_cnb, s_cpiece[4], s_cmax[2], s_fx[3], s_fxnb, s_fxscale, s_ckind;
	if constexpr (W == 4) {
		if (allow_compact) {
			const u32 raw0 = (u32)src[0];
			u32 raw[S];
			u32 v = 0;
#pragma unroll
			for (u32 e = 0; e < S; ++e) {
				raw[e] = (u32)src[i0 + e];
				v |= raw[e] ^ raw0;
			}
#pragma unroll
			for (int off = 32; off > 0; off >>= 1)
				v |= (u32)__shfl_xor((int)v, off);
			if (tid == 0)
				s_vraw = s_cnb = s_cmax[0] = s_cmax[1] = s_ckind = s_fxnb = s_fxscale = 0;
			if (tid < 512)
				(&hs[0][0])[tid] = 0;
			__syncthreads();
			if ((tid & 63) == 0)
				atomicOr(&s_vraw, v);
			__syncthreads();
			if (tid == 0) {
				// the runs of varying bits, highest first; their packed places, highest first
				u32 vv = s_vraw, np = 0, nb = (u32)__popc(vv), left = nb;
				bool ok = nb >= 17u && nb <= 30u;
				u32 pc[4] = {0, 0, 0, 0};
				while (vv && ok) {
					const u32 hi = 31u - (u32)__builtin_clz(vv);            // top bit of the highest run
					const u32 below = ~vv & ((2u << hi) - 1u);              // the bits below it that do NOT vary
					const u32 lo = below ? 32u - (u32)__builtin_clz(below) : 0u;   // the run is bits lo .. hi
					const u32 w = hi - lo + 1u;
					if (np == 4) {
						ok = false;
						break;
					}
					left -= w;
					pc[np++] = lo | (w << 8) | (left << 16);
					vv &= lo ? (1u << lo) - 1u : 0u;
				}
				for (u32 i = 0; i < 4; ++i)
					s_cpiece[i] = pc[i];
				s_cnb = ok ? nb : 0u;
			}
			__syncthreads();
			const u32 cnb = s_cnb;
			if (cnb) {
				const u32 pc[4] = {s_cpiece[0], s_cpiece[1], s_cpiece[2], s_cpiece[3]};
#pragma unroll
				for (u32 e = 0; e < S; ++e) {
					const u32 c = compact_key<KT>(raw[e], pc, cnb, ka);
					atomicAdd(&hs[0][(c >> (cnb - 8u)) & 0xFFu], 1u);
					atomicAdd(&hs[1][(c >> (cnb - 16u)) & 0xFFu], 1u);
				}
			}
			__syncthreads();
			if (cnb) {
				u32 ms = tid < 512 ? (&hs[0][0])[tid] : 0u;
#pragma unroll
				for (int off = 32; off > 0; off >>= 1) {
					const u32 ys = __shfl_xor(ms, off);
					ms = ys > ms ? ys : ms;
				}
				if ((tid & 63) == 0 && tid < 512)
					atomicMax(&s_cmax[tid >> 8], ms);
			}
			__syncthreads();
			// ---- floats on a grid (SegCtl::ckind == 1), where the packed bits do not spread
			const bool packed_ok = s_cnb && s_cmax[0] <= 2 * NS / 256 && s_cmax[1] <= 2 * NS / 256;
			if (!packed_ok && ka.fmask != 0) {
				u32 lo = 0xFFFFFFFFu, hi = 0, flags = 0;   // lowest set bit's weight / magnitude bound (both + 256), not finite | -0.0
#pragma unroll
				for (u32 e = 0; e < S; ++e) {
					const u32 E = (raw[e] >> 23) & 0xFFu, M = raw[e] & 0x7FFFFFu;
					flags |= (E == 255u || raw[e] == 0x80000000u) ? 1u : 0u;
					flags |= (raw[e] >> 31) ? 2u : 0u;
					if (raw[e] & 0x7FFFFFFFu) {
						const u32 mant = E ? (M | 0x800000u) : M, Ee = E ? E : 1u;
						const u32 w = 256u + Ee - 150u + (u32)__builtin_ctz(mant);
						lo = w < lo ? w : lo;
						const u32 t = 256u + Ee - 126u;
						hi = t > hi ? t : hi;
					}
				}
#pragma unroll
				for (int off = 32; off > 0; off >>= 1) {
					const u32 a = (u32)__shfl_xor((int)lo, off), b = (u32)__shfl_xor((int)hi, off);
					lo = a < lo ? a : lo;
					hi = b > hi ? b : hi;
					flags |= (u32)__shfl_xor((int)flags, off);
				}
				if (tid == 0) {
					s_fx[0] = 0xFFFFFFFFu;
					s_fx[1] = s_fx[2] = 0;
					s_cmax[0] = s_cmax[1] = 0;
					s_fxnb = 0;
				}
				if (tid < 512)
					(&hs[0][0])[tid] = 0;
				__syncthreads();
				if ((tid & 63) == 0) {
					atomicMin(&s_fx[0], lo);
					atomicMax(&s_fx[1], hi);
					atomicOr(&s_fx[2], flags);
				}
				__syncthreads();
				if (tid == 0 && (s_fx[2] & 1u) == 0 && s_fx[1] > s_fx[0]) {
					const int e0 = (int)s_fx[0] - 256, e1 = (int)s_fx[1] - 256;
					const int nb = e1 - e0 + ((s_fx[2] & 2u) ? 1 : 0);   // (a sign bit only if some key is negative)
					if (nb >= 17 && nb <= 30 && 127 - e0 >= 1 && 127 - e0 <= 254) {
						s_fxnb = (u32)nb;
						s_fxscale = (u32)(127 - e0) << 23;
					}
				}
				__syncthreads();
				const u32 fnb = s_fxnb;
				if (fnb) {
					const u32 sc = s_fxscale;
					u32 bad = 0;
#pragma unroll
					for (u32 e = 0; e < S; ++e) {
						const u32 c = fixedpoint_key(raw[e], sc, fnb, (s_fx[2] & 2u) == 0, ka.desc != 0, bad);
						atomicAdd(&hs[0][(c >> (fnb - 8u)) & 0xFFu], 1u);
						atomicAdd(&hs[1][(c >> (fnb - 16u)) & 0xFFu], 1u);
					}
					(void)bad;   // (the sample's own keys lie on the grid it was made from)
				}
				__syncthreads();
				if (fnb) {
					u32 ms = tid < 512 ? (&hs[0][0])[tid] : 0u;
#pragma unroll
					for (int off = 32; off > 0; off >>= 1) {
						const u32 ys = __shfl_xor(ms, off);
						ms = ys > ms ? ys : ms;
					}
					if ((tid & 63) == 0 && tid < 512)
						atomicMax(&s_cmax[tid >> 8], ms);
				}
				__syncthreads();
				if (tid == 0 && fnb) {
					s_cnb = fnb;           // (the decision below reads s_cnb / s_cmax: now the grid's)
					s_ckind = 1;
				}
				__syncthreads();
			}
		}
	}
	if (tid == 0) {
		bool go = s_desc != 0;
		u32 nk = 0, cols[8] = {0, 0, 0, 0, 0, 0, 0, 0};
		u64 cmask = 0;
		for (u32 c = 0; c < W; ++c) {
			if (s_distinct[c] >= 2)
				cols[nk++] = c;                      // proved kept
			else
				cmask |= (u64)0xFFu << (8 * c);      // taken for constant; the level-1 pass will know
		}
		if (shift1 + 8 < 8 * W)                      // ... and so are the bits above the level-1 digit
			cmask |= ~(((u64)1 << (shift1 + 8)) - 1) & (W == 8 ? ~(u64)0 : (((u64)1 << (8 * (W & 7))) - 1));
		go = go && nk >= min_cols;
		for (u32 i = 0; i + 2 < nk; ++i) {
			// the columns the leaves sort by: no digit with a tenth of the sample (Plan::hot: lanes queue at one counter)
			go = go && s_max[cols[i]] <= NS / 10;
		}
		// the two digits the MSB passes go by (at their bit positions: the two highest kept columns, or below the highest
		// varying bit): none with twice its share of the sample (a slot holds 1.25 times the mean)
		// (hints & 1: the level-1 digit's test is the caller's -- keys that an MSD split has ordered by their top byte, piece by piece,
		// are even over the array and clustered at every one of the sample's 64 places: the local sorts of a distributed sort)
		go = go && ((hints & 1u) || s_maxs[0] <= 2 * NS / 256) && s_maxs[1] <= 2 * NS / 256;
		// (digits that are not bytes: the leaves of round 3, which take over when the bins are uneven, sort by bytes)
		// the leaves' bins (4096 of them, fewer when fewer than twelve bits are left below the MSB digits): NS samples give each
		// NS / bins on average; a bin with three times that (and a margin for the small counts) means clustered low bits
		const u32 nbins12 = (shift1 & 7u) && shift2 < 12u ? 1u << shift2 : 4096u;
		const u32 max12_ok = 3 * (NS / nbins12) + 18u;
		if (shift1 & 7u)
			go = go && s_max12 <= max12_ok;
		// the byte scheme does not take these keys, their packed varying bits would (all four columns kept, so that the ranks end in
		// the half the reference's parity rule names for four passes, radix_sort_rank.hpp:91; unsorted; both packed MSB digits even)
		u32 compact = 0;
		if constexpr (W == 4) {
			if (!go && allow_compact && s_cnb && s_desc != 0 && nk == 4 && s_cmax[0] <= 2 * NS / 256 && s_cmax[1] <= 2 * NS / 256) {
				compact = s_cnb;
				go = true;
				shift1 = compact - 8u;
				shift2 = compact - 16u;
				cmask = 0;   // (what must not vary is checked on the keys as the caller wrote them: cvnot)
			}
		}
		if (backoff && ctl->boff_skip) {   // (an attempt of this context was lost lately: SegCtl::boff_skip)
			ctl->boff_skip -= 1u;
			go = false;
			compact = 0;
		}
		ctl->compact = compact;
		ctl->ckind = compact ? s_ckind : 0u;
		for (u32 i = 0; i < 4; ++i)
			ctl->cpiece[i] = compact ? (s_ckind ? (i == 0 ? s_fxscale : i == 1 ? ((s_fx[2] & 2u) ? 0u : 1u) : 0u) : s_cpiece[i]) : 0u;   // (grid: the factor, "no key is negative")
		ctl->craw0 = compact ? (u32)src[0] : 0u;
		ctl->cvnot = compact ? ~s_vraw : 0u;
		ctl->ntiles = ctl->mode = ctl->maxleaf = ctl->done = ctl->nleaf = ctl->overflow = ctl->nredo = 0;   // (nobody else zeroes the control block)
		ctl->blind = go ? BLIND_GO : BLIND_FAILED;
		// NS samples over 4096 bins: two per bin on average, the fullest holds ten or eleven; a leaf of 4096 keys sees half of
		// what the sample sees, and rsx_leaf16_kernel takes bins of up to 25 keys
		ctl->leaf16 = (s_max12 <= max12_ok || compact) ? 1u : 0u;   // (packed keys: the leaves' own test of their bins decides)
		// (8-byte keys, the leaves' columns all in the low word, their bins even: the leaves that read four-byte slots)
		// ... 2 (allow_narrow >= 2: the caller has enqueued that form of the level-1 pass too): nothing below the LEVEL-1 digit varies
		// above bit 32 either -- the level-1 slots hold low words already (rsx_pass32a_kernel, OT = u32)
		u32 narrow = (W == 8 && allow_narrow && go && nk >= 4 && cols[nk - 3] <= 3u && s_max12 <= max12_ok) ? 1u : 0u;
		if (narrow && allow_narrow >= 2u && ((shift1 & 7u) ? shift1 <= 32u : cols[nk - 2] <= 3u))
			narrow = 2u;
		ctl->narrow = narrow;
		ctl->shift1 = shift1;
		ctl->shift2 = shift2;
		ctl->cmask_lo = (u32)cmask;
		ctl->cmask_hi = (u32)(cmask >> 32);
		ctl->key0_lo = (u32)(u64)k[0];               // (thread 0's first sample is the array's first key)
		ctl->key0_hi = (u32)((u64)k[0] >> 32);
		if (go) {
			Plan *const out[2] = {plan, host_plan};
			for (int j = 0; j < 2; ++j) {
				out[j]->ncols = nk;
				out[j]->sorted = 0;
				for (u32 i = 0; i < 8; ++i)
					out[j]->cols[i] = i < nk ? cols[i] : 0u;
				out[j]->hot = 0;
				out[j]->vary_lo = out[j]->vary_hi = 0;
				out[j]->hyb = HYB_TWO_LEVEL;
				out[j]->max1 = 0;
			}
		}
	}
}

// ---- the leaves ---------------------------------------------------------------------------------------------------------
// A workgroup sorts a leaf at a time: the leaf's keys go into registers (wave w owns a contiguous slice, round r of lane l is
// the slice's element 64 r + l: memory order, as in rsx_scatter2_kernel), then per remaining kept column, LSB first: a
// returning LDS atomic on the (wave, digit) cell is the key's rank in its run, the cells become run starts, the key is staged
// at start + rank, and the slice is read back for the next column.  After the last column the staged leaf is written out in
// 16-byte pieces.  A workgroup takes leaf s, s + grid, ... of the table; the launches of the level-2 leaves give every table
// entry a workgroup of its own (several fit a CU, so its loads overlap the others' sorting: 0.569 against 0.585 ms for 2^28
// keys with 8192 persistent workgroups).  PREFETCH_ (the next leaf's keys requested before the current one is sorted) is
// kept for the probe; no shipped shape uses it.  Rests on the LDS resolving same-address lanes of a returning atomic in lane
// order, as the pass kernel does (checked on the device before either is used: lds_order_selfcheck, rsx.hip).
template <typename KT, int NW_, int KPT_, int WPE_ = 1, bool RANK1_ = true, bool PREFETCH_ = false> struct LeafCfg {
	static constexpr int NW = NW_, KPT = KPT_, BLOCK = NW_ * 64, CAP = NW_ * 64 * KPT_;
	static constexpr int WPE = WPE_;   // waves per SIMD the register allocation must leave room for
	static constexpr bool RANK1 = RANK1_;       // one returning atomic per key and column (its rank, kept in 16 bits) or two
	static constexpr bool PREFETCH = PREFETCH_; // the next leaf's keys requested before the current one is sorted
	static_assert(CAP <= 32768, "ranks and positions are kept in 16 bits");
};

// level 1: the 256 buckets of the highest kept column, bounds from its offsets; level 2: segtab[0 .. ctl->nleaf).
// Enqueued before the host knows what the plan says: does nothing unless the plan is `level` and the largest leaf lies in
// (lo, hi] -- the host launches the shape for small leaves and, where it may be needed, the one that fills the LDS; the
// device picks.  Reads the buffer the last pass wrote (level 1: aux, level 2: src) and writes where plan.ncols passes would
// end (radix_sort.hpp:92).
// CT: the type a key is CARRIED in inside the leaf.  All keys of a leaf agree in the columns above the ones it sorts by (the
// MSB passes' digits; skipped columns are the same in every key), so when those columns all lie in the low half of the key
// -- 8-byte keys whose top bytes are zero, BASELINE.json's cfg 3 -- registers and LDS hold 4 bytes per key and the upper
// half is put back when the keys leave (one copy per leaf): the leaf of 2^28 u64 keys with three low columns then costs what
// a u32 leaf does instead of three times as much.  The instantiation with CT narrower than KT takes exactly the leaves that
// allow it; `skip_narrowable` tells the KT-wide instantiation launched beside it to leave those alone.
// DENSE (with CT narrower than KT): the slots hold CT elements -- the low bytes of the DERIVED keys, written by a level-2 pass
// with KTO = CT (rsx_scatter2.hpp) -- and nothing else of the keys exists there: the bytes above CT are the two MSB digits the
// slot stands for and, in columns that were skipped, the bytes of the array's first key (SegCtl::key0; sorts without a
// histogram only).  Half the bytes written by the pass and read here for 4-byte keys with two columns per leaf.
template <typename KT, typename C, typename CT = KT, bool DENSE = false>
__global__ __launch_bounds__(C::BLOCK, C::WPE) void rsx_leaf_sort_kernel(KT *__restrict__ src, KT *__restrict__ aux, u64 n,
                                                                  const u64 *__restrict__ ghist, const Plan *__restrict__ plan,
                                                                  const LeafSeg *__restrict__ segtab,
                                                                  const SegCtl *__restrict__ ctl, KdfArgs<KT> ka, u32 level,
                                                                  u32 lo, u32 hi, const KT *__restrict__ slots = nullptr,
                                                                  u32 slack_cap = 0, u32 skip_narrowable = 0,
                                                                  const u64 *__restrict__ off1_given = nullptr,
                                                                  const u32 *__restrict__ redo = nullptr)
{
	constexpr int NW = C::NW, KPT = C::KPT, BLOCK = C::BLOCK;
	constexpr bool NARROW = sizeof(CT) < sizeof(KT);
	constexpr int CHUNK = 16 / sizeof(CT);
	// rounds per guarded group (a wave's slice is a whole number of groups).  Groups of two rounds balance the waves of a
	// 4100-key leaf better (18 instead of 20 rounds in the longest wave) and change nothing: 0.585 against 0.589 ms in the
	// 20-round shape, single rounds 0.630 (tools/ubench/leaf_probe); the 32-round shapes spill with either.
	constexpr int G = 4;
	static_assert(KPT % G == 0, "whole groups of rounds");
	// everything the decision needs is requested at once (scalar loads), not one dependent round trip after the other
	const u32 hyb = plan->hyb, ncols = plan->ncols, max1 = plan->max1;
	u32 colpack = 0;   // 4 bits per kept column
#pragma unroll
	for (int k = 0; k < 8; ++k)
		colpack |= (plan->cols[k] & 15u) << (4 * k);
	const u32 mode = level == HYB_TWO_LEVEL ? ctl->mode : (u32)SEG_MODE_LEAVES;
	const u32 maxleaf = level == HYB_TWO_LEVEL ? ctl->maxleaf : max1;
	// redo: the launch behind rsx_leaf16_kernel (rsx_leaf16.hpp) -- only the table entries its list names (SegCtl::nredo of
	// them), or every entry if that kernel was told to stay away (SegCtl::leaf16 == 0)
	const bool listed = redo != nullptr && ctl->leaf16 != 0;
	const u32 nseg = level == HYB_TWO_LEVEL ? (listed ? ctl->nredo : ctl->nleaf) : 256u;
	if (hyb != level || mode != SEG_MODE_LEAVES || maxleaf <= lo || maxleaf > hi)
		return;
	// the level-1 buckets' starts: the highest kept column's scanned offsets (from a self-planned pass 0: its own copy)
	const u64 *off1 = off1_given ? off1_given : ghist + 256 * ((colpack >> (4 * (ncols - 1))) & 15u);
	const KT *in = level == HYB_TWO_LEVEL ? src : aux;
	KT *out = (ncols & 1) ? aux : src;

	__shared__ __attribute__((aligned(16))) CT stage[C::CAP];
	__shared__ u32 cell[NW][256];
	__shared__ u32 wsum[4];
	__shared__ KT s_upper;   // NARROW: what every key of the current leaf has above its carried bits (derived key)
	const u32 tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
	auto opaque = [](u32 x) {
		asm volatile("" : "+v"(x));
		return x;
	};
	// A wave's slice is a whole number of groups of G rounds; what lies behind the leaf's end is padded with keys whose
	// derived key is all ones: they have digit 255 in every column and come last in memory order, so every (stable) pass
	// leaves them behind the leaf's keys -- no lane ever tests whether its element exists.
	// Inside a leaf the keys live in registers and LDS as their DERIVED keys (kdf_apply once when they arrive, kdf_invert once
	// when they leave: the element images that reach memory are the caller's, bit for bit), so that a digit is one bit-field
	// extract; with the KDF's arithmetic per column and phase the leaves were bound by vector instructions, not by the LDS.
	const CT pad = (CT)~(CT)0;
	auto bounds = [&](u32 s, u32 &beg, u32 &cnt, u32 &nc, u32 &slot) {
		slot = 0;
		if (level == HYB_TWO_LEVEL) {
			const LeafSeg ls = segtab[listed ? redo[s] : s];
			beg = ls.beg;
			cnt = ls.cnt;
			nc = ls.ncols & 0xFFFFu;   // (the upper half: how many of the slot's values lie at its end, rsx_leaf16_kernel)
			slot = ls.slot;
		} else {
			const u64 b = off1[s], e = s == 255 ? n : off1[s + 1];
			beg = (u32)b;
			cnt = (u32)(e - b);
			nc = ncols - 1;
		}
	};
	// groups of rounds of this wave that hold any of the leaf's keys: what lies wholly behind the leaf's end is not touched
	// (a round of nothing but padding is 64 lanes on ONE counter: the slowest thing an LDS atomic can be asked to do)
	const u32 swid = (u32)__builtin_amdgcn_readfirstlane((int)wid);
	auto wave_groups = [&](u32 cnt, u32 ng) {
		const u32 first = swid * ng * (64 * G);
		const u32 mine = cnt > first ? cnt - first : 0u;
		const u32 g = (mine + 64 * G - 1) / (64 * G);
		return g < ng ? g : ng;
	};
	// The keys arrive as the caller's images: request() only ISSUES the loads (all of a leaf's groups in flight together; a
	// first version converted group by group and so waited for every group's loads before requesting the next: 0.63 instead
	// of 0.56 ms for the leaves of 2^28 keys), derive() turns what has arrived into derived keys cut to the carried type.
	// NARROW: only the low half of every key is read -- all keys of such a leaf agree in the upper half, of the image as of
	// the derived key (the KDF flips bits by the image's top bit only) -- and the leaf's first key gives the upper half.
	KT first_raw = 0;
	auto request = [&](auto &dst, u32 beg, u32 cnt, u32 slot) {
		const u32 ngall = (cnt + BLOCK * G - 1) / (BLOCK * G);
		const u32 ng = wave_groups(cnt, ngall);
		const u32 wo = opaque(wid * (ngall * 64 * G) + lane);
		if constexpr (DENSE) {
			const CT *q = (const CT *)slots + (u64)(slot - 1) * slack_cap;
#pragma unroll
			for (int g = 0; g < KPT / G; ++g) {
				if (g < (int)ng) {
#pragma unroll
					for (int r = g * G; r < (g + 1) * G; ++r) {
						const u32 i = wo + r * 64;
						dst[r] = i < cnt ? q[i] : pad;
					}
				}
			}
			return;
		}
		const KT *p = slot ? slots + (u64)(slot - 1) * slack_cap : in + beg;
		if constexpr (NARROW)
			first_raw = p[0];
#pragma unroll
		for (int g = 0; g < KPT / G; ++g) {
			if (g < (int)ng) {
#pragma unroll
				for (int r = g * G; r < (g + 1) * G; ++r) {
					const u32 i = wo + r * 64;
					if constexpr (NARROW)
						dst[r] = i < cnt ? ((const CT *)p)[(sizeof(KT) / sizeof(CT)) * i] : (CT)kdf_invert((KT)~(KT)0, ka);
					else
						dst[r] = i < cnt ? p[i] : kdf_invert((KT)~(KT)0, ka);
				}
			}
		}
	};
	auto derive = [&](auto &dst, u32 cnt, u32 slot) {
		if constexpr (DENSE) {
			// the slots hold derived keys already; what lies above the carried bytes follows from the slot
			if (tid == 0) {
				const u32 c1 = (colpack >> (4 * (ncols - 1))) & 15u, c2 = (colpack >> (4 * (ncols - 2))) & 15u;
				const KT key0 = (KT)(((u64)ctl->key0_hi << 32) | ctl->key0_lo);
				const KT digits = (KT)((KT)0xFFu << (8 * c1)) | (KT)((KT)0xFFu << (8 * c2));
				const KT low = (KT)(((KT)1 << (8 * sizeof(CT))) - 1);
				s_upper = (KT)((key0 & ~digits & ~low) | ((KT)((slot - 1) >> 8) << (8 * c1)) | ((KT)((slot - 1) & 255u) << (8 * c2)));
			}
			return;
		}
		// (straight-line over all rounds, also those of groups that were not requested: the compiler can then wait for the
		// loads one by one instead of for all of them at the first group's door; what it derives from a register that was
		// never loaded is never looked at)
		const u32 ngall = (cnt + BLOCK * G - 1) / (BLOCK * G);
		const u32 wo = opaque(wid * (ngall * 64 * G) + lane);
		constexpr u32 CBITS = NARROW ? 8 * sizeof(CT) : 0;   // (0: nothing above the carried bits)
		KT upper_raw = 0;
		if constexpr (NARROW) {
			upper_raw = (KT)(first_raw >> CBITS << CBITS);
			if (tid == 0)   // what every derived key of this leaf has above its carried bits
				s_upper = (KT)(kdf_apply(first_raw, ka) >> CBITS << CBITS);
		}
#pragma unroll
		for (int r = 0; r < KPT; ++r) {
			if constexpr (NARROW)   // (the padding's low half must be all ones AFTER the derivation with this leaf's upper half)
				dst[r] = wo + r * 64 < cnt ? (CT)kdf_apply((KT)(upper_raw | (KT)dst[r]), ka) : pad;
			else
				dst[r] = (CT)kdf_apply((KT)dst[r], ka);
		}
	};
	u32 s = blockIdx.x;
	if (s >= nseg)
		return;
	u32 nbeg, ncnt, nnc, nslot;
	bounds(s, nbeg, ncnt, nnc, nslot);
	static_assert(!C::PREFETCH || !NARROW, "one upper part at a time");
	CT nxt[C::PREFETCH ? KPT : 1];
	if constexpr (C::PREFETCH)
		request(nxt, nbeg, ncnt, nslot);
	for (;;) {
		const u32 beg = nbeg, cnt = ncnt, nrem = nnc, slot = nslot;
		// is this leaf this instantiation's?  (all its columns -- ascending -- inside the carried type, or not)
		const bool narrowable = nrem != 0 && ((colpack >> (4 * (nrem - 1))) & 15u) < 4u && sizeof(KT) == 8;
		if (cnt == 0 || (!DENSE && ((NARROW && !narrowable) || (!NARROW && (skip_narrowable & 1u) && narrowable)))) {   // (cnt 0: an empty bucket's table entry)
			s += gridDim.x;
			if (s >= nseg)
				break;
			bounds(s, nbeg, ncnt, nnc, nslot);
			continue;
		}
		const u32 ngall = (cnt + BLOCK * G - 1) / (BLOCK * G);   // groups of rounds in a wave's slice
		const u32 per = ngall * (64 * G);
		const u32 ng = wave_groups(cnt, ngall);                  // ... and those this wave has keys in (wave-uniform)
		const u32 wo0 = wid * per + lane;
		CT keep[KPT];
		if constexpr (C::PREFETCH) {
#pragma unroll
			for (int r = 0; r < KPT; ++r)
				keep[r] = nxt[r];
		} else {
			request(keep, beg, cnt, slot);
		}
		derive(keep, cnt, slot);
		s += gridDim.x;
		const bool more = s < nseg;
		if (more) {
			bounds(s, nbeg, ncnt, nnc, nslot);
			if constexpr (C::PREFETCH)
				request(nxt, nbeg, ncnt, nslot);
		}
		// 8-byte keys with five or more columns left (BASELINE.json's cfg 3, all eight columns kept: six per leaf).  Keys-only,
		// so equal keys are the same bits and ANY sorted order is the reference's output: the leaf is sorted by its TOP three
		// columns only (LSB first among them), after which evenly spread keys are in order but for the few that share those
		// 24 bits -- 4096 keys: half a pair per leaf -- and odd-even transposition on whole keys puts those right in one
		// sweep and finds nothing to do in the next.  Keys that cluster in these columns (the sweeps do not come to rest
		// within four rounds) get what every leaf got before: all columns, LSB first, from wherever the keys lie now.
		// (2^28 u64 keys, six columns per leaf, one box: the sort takes 3.79 instead of 4.55 ms; with the top TWO columns and
		// the three sweeps those need: 3.87 against 4.58 -- a sweep costs more than a third of a column.)
		constexpr bool PREFIX_OK = sizeof(CT) == 8;
		u32 c0 = 0;
		if constexpr (PREFIX_OK) {
			if (nrem >= 5 && !(skip_narrowable & 2u))
				c0 = nrem - 3;
		}
		for (;;) {
		for (u32 c = c0; c < nrem; ++c) {
			const u32 shift = 8 * ((colpack >> (4 * c)) & 15u);
#pragma unroll
			for (int k = 0; k < 4; ++k)
				cell[wid][lane + 64 * k] = 0;
			// (a wave's DS operations execute in order: no barrier between zeroing and counting its own row)
			u32 *wc = cell[wid];
			u32 rk[C::RANK1 ? KPT / 2 : 1];
			if constexpr (C::RANK1) {
#pragma unroll
				for (int i = 0; i < KPT / 2; ++i)
					rk[i] = 0;
			}
#pragma unroll
			for (int g = 0; g < KPT / G; ++g) {
				if (g < (int)ng) {
#pragma unroll
					for (int r = g * G; r < (g + 1) * G; ++r) {
						const u32 old = __hip_atomic_fetch_add(&wc[(u32)(keep[r] >> shift) & 0xFFu], 1u, __ATOMIC_RELAXED,
						                                       __HIP_MEMORY_SCOPE_WORKGROUP);
						if constexpr (C::RANK1)
							rk[r >> 1] |= old << (16 * (r & 1));   // the key's rank in its (wave, digit) run
					}
				}
			}
			__syncthreads();
			u32 tot = 0, incl = 0;
			if (tid < 256) {
#pragma unroll
				for (int w = 0; w < NW; ++w)
					tot += cell[w][tid];
				u32 x = tot;
#pragma unroll
				for (int off = 1; off < 64; off <<= 1) {
					const u32 y = __shfl_up(x, off);
					if (lane >= (u32)off)
						x += y;
				}
				incl = x;
				if (lane == 63)
					wsum[wid] = x;
			}
			__syncthreads();
			if (tid < 256) {
				u32 acc = incl - tot;   // radix_sort.hpp:72-80, the exclusive scan over the digits
				for (u32 w = 0; w < wid; ++w)
					acc += wsum[w];
#pragma unroll
				for (int w = 0; w < NW; ++w) {
					const u32 k = cell[w][tid];
					cell[w][tid] = acc;
					acc += k;
				}
			}
			__syncthreads();
#pragma unroll
			for (int g = 0; g < KPT / G; ++g) {
				if (g < (int)ng) {
					// the key's place: its run's start + its rank (or the returning atomic on the run's cursor): rounds in memory
					// order, lanes in lane order
					u32 pos[G];
#pragma unroll
					for (int r = g * G; r < (g + 1) * G; ++r) {
						if constexpr (C::RANK1)
							pos[r - g * G] = wc[(u32)(keep[r] >> shift) & 0xFFu] + ((rk[r >> 1] >> (16 * (r & 1))) & 0xFFFFu);
						else
							pos[r - g * G] = __hip_atomic_fetch_add(&wc[(u32)(keep[r] >> shift) & 0xFFu], 1u, __ATOMIC_RELAXED,
							                                        __HIP_MEMORY_SCOPE_WORKGROUP);
					}
#pragma unroll
					for (int r = g * G; r < (g + 1) * G; ++r)
						stage[pos[r - g * G]] = keep[r];
				}
			}
			__syncthreads();
			if (c + 1 < nrem) {
				const u32 wo = opaque(wo0);
#pragma unroll
				for (int g = 0; g < KPT / G; ++g) {
					if (g < (int)ng) {
#pragma unroll
						for (int r = g * G; r < (g + 1) * G; ++r)
							keep[r] = stage[wo + r * 64];
					}
				}
				// (the next column stages only behind two more barriers: every slice has been read back by then)
			}
		}
		if constexpr (!PREFIX_OK) {
			break;
		} else {
			if (c0 == 0)
				break;
			bool rest = false;
			for (u32 it = 0; it < 4 && !rest; ++it) {
				u32 sw = 0;
				for (u32 i = 2 * tid; i + 1 < cnt; i += 2 * BLOCK) {          // pairs (2 i, 2 i + 1)
					const CT a = stage[i], b = stage[i + 1];
					if (a > b) {
						stage[i] = b;
						stage[i + 1] = a;
						sw = 1;
					}
				}
				__syncthreads();
				for (u32 i = 2 * tid + 1; i + 1 < cnt; i += 2 * BLOCK) {      // pairs (2 i + 1, 2 i + 2)
					const CT a = stage[i], b = stage[i + 1];
					if (a > b) {
						stage[i] = b;
						stage[i + 1] = a;
						sw = 1;
					}
				}
				rest = __syncthreads_or((int)sw) == 0;
			}
			if (rest)
				break;
			c0 = 0;   // clustered after all: every column
			const u32 wo = opaque(wo0);
#pragma unroll
			for (int g = 0; g < KPT / G; ++g) {
				if (g < (int)ng) {
#pragma unroll
					for (int r = g * G; r < (g + 1) * G; ++r)
						keep[r] = stage[wo + r * 64];
				}
			}
			__syncthreads();
		}
		}
		// write out: 16 bytes per lane (the leaf's start is only element-aligned)
		if (nrem) {
			KT *o = out + beg;
			for (u32 i0 = tid * CHUNK; i0 < cnt; i0 += BLOCK * CHUNK) {
				typedef CT kvec_t __attribute__((ext_vector_type(CHUNK)));
				const kvec_t x = *(const kvec_t *)&stage[i0];
				KT kv[CHUNK];
#pragma unroll
				for (int e = 0; e < CHUNK; ++e)
					kv[e] = kdf_invert(NARROW ? (KT)(s_upper | (KT)x[e]) : (KT)x[e], ka);
				if (i0 + CHUNK <= cnt) {
					if constexpr (sizeof(KT) * CHUNK > 16) {   // (carried narrower than stored: two 16-byte stores)
						KT lo2[CHUNK / 2], hi2[CHUNK / 2];
#pragma unroll
						for (int e = 0; e < CHUNK / 2; ++e) {
							lo2[e] = kv[e];
							hi2[e] = kv[CHUNK / 2 + e];
						}
						store_chunk<KT, CHUNK / 2>(o + i0, lo2);
						store_chunk<KT, CHUNK / 2>(o + i0 + CHUNK / 2, hi2);
					} else {
						store_chunk<KT, CHUNK>(o + i0, kv);
					}
				} else {
#pragma unroll
					for (int e = 0; e < CHUNK; ++e)
						if (i0 + e < cnt)
							o[i0 + e] = kv[e];
				}
			}
		}
		if (!more)
			break;
		__syncthreads();   // the staged leaf has been read before the next one is staged
	}
}

// ---- leaves of key + payload sorts (4-byte keys, 4-byte payloads: BASELINE.json's cfg 4) ------------------------------------
// The slack route of a two-level sort for (key, payload) pairs and for rank sorts (payload = the element's index,
// radix_sort_rank.hpp): two MSB passes of the key + payload pass kernel, the second into per-bucket slots of two scratch
// arrays, then this kernel -- rsx_leaf_sort_kernel's algorithm with the pair carried as ONE 8-byte value, derived key in the
// upper half (so a key column c is byte 4 + c of what is ranked) and the payload in the lower.  A leaf gathers from its slot
// and writes the payloads (and, for pair sorts, the keys) to its place in the dense result.  Rank sorts do not want the keys.
// level == HYB_ONE_LEVEL (mid-size arrays): the 256 buckets of ONE pass by the highest kept column, read where that pass wrote
// them (kslots / vslots are then its dense output, the bounds the column's offsets in `ghist`).
// K16 (two levels, sorts without a histogram): the key slots hold the low two bytes of the derived (or packed) keys
// (rsx_leafp_kernel, rsx_leaf16.hpp); the two bytes above them are the slot's two digits.
template <typename KT, typename VT, typename C, bool K16 = false>
__global__ __launch_bounds__(C::BLOCK, C::WPE) void rsx_leaf_pairs_kernel(const KT *__restrict__ kslots, const VT *__restrict__ vslots,
                                                                           u32 slack_cap, KT *__restrict__ kout, VT *__restrict__ vout,
                                                                           const Plan *__restrict__ plan,
                                                                           const LeafSeg *__restrict__ segtab,
                                                                           const SegCtl *__restrict__ ctl, KdfArgs<KT> ka,
                                                                           u32 level = HYB_TWO_LEVEL,
                                                                           const u64 *__restrict__ ghist = nullptr, u64 n = 0,
                                                                           const u32 *__restrict__ redo = nullptr)
{
	static_assert(sizeof(KT) == 4 && sizeof(VT) == 4, "pairs of 4-byte keys and 4-byte payloads");
	constexpr int NW = C::NW, KPT = C::KPT, BLOCK = C::BLOCK, G = 4;
	static_assert(KPT % G == 0, "whole groups of rounds");
	if (plan->hyb != level || (level == HYB_TWO_LEVEL && ctl->mode != SEG_MODE_LEAVES))
		return;
	if (level == HYB_TWO_LEVEL && ctl->compact)
		ka.fmask = ka.sflip = ka.desc = 0;   // (the slots hold packed keys, SegCtl::compact: plain unsigned)
	u32 colpack = 0;
#pragma unroll
	for (int k = 0; k < 8; ++k)
		colpack |= (plan->cols[k] & 15u) << (4 * k);
	const u32 ncols_all = plan->ncols;
	// redo: the launch behind rsx_leafp_kernel (rsx_leaf16.hpp) -- the leaves its list names, or all of them (SegCtl::leaf16 == 0)
	const bool listed = redo != nullptr && ctl->leaf16 != 0;
	const u32 nseg = level == HYB_TWO_LEVEL ? (listed ? ctl->nredo : ctl->nleaf) : 256u;
	const u64 *off1 = ghist + 256 * ((colpack >> (4 * (ncols_all - 1))) & 15u);
	__shared__ __attribute__((aligned(16))) u64 stage[C::CAP];
	__shared__ u32 cell[NW][256];
	__shared__ u32 wsum[4];
	const u32 tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
	const u32 swid = (u32)__builtin_amdgcn_readfirstlane((int)wid);
	auto opaque = [](u32 x) {
		asm volatile("" : "+v"(x));
		return x;
	};
	for (u32 s = blockIdx.x; s < nseg; s += gridDim.x) {
		LeafSeg ls;
		if (level == HYB_TWO_LEVEL) {
			ls = segtab[listed ? redo[s] : s];
		} else {
			const u64 b = off1[s], e = s == 255 ? n : off1[s + 1];
			ls.beg = (u32)b;
			ls.cnt = (u32)(e - b);
			ls.ncols = ncols_all - 1;
			ls.slot = 0;
		}
		const u32 cnt = ls.cnt, nrem = ls.ncols;
		if (cnt == 0)
			continue;
		const u32 ngall = (cnt + BLOCK * G - 1) / (BLOCK * G);
		const u32 per = ngall * (64 * G);
		const u32 first = swid * per;
		const u32 mine = cnt > first ? cnt - first : 0u;
		u32 ng = (mine + 64 * G - 1) / (64 * G);
		ng = ng < ngall ? ng : ngall;
		const u32 wo0 = wid * per + lane;
		const KT *kp = ls.slot ? kslots + (u64)(ls.slot - 1) * slack_cap : kslots + ls.beg;
		const unsigned short *kp16 = (const unsigned short *)kslots + (u64)(ls.slot ? ls.slot - 1 : 0u) * slack_cap;
		const KT upper16 = (KT)(((KT)((ls.slot - 1) >> 8) << 24) | ((KT)((ls.slot - 1) & 255u) << 16));
		const VT *vp = ls.slot ? vslots + (u64)(ls.slot - 1) * slack_cap : vslots + ls.beg;
		KT kr[KPT];   // (K16: derived keys)
		VT vr[KPT];
		{
			const u32 wo = opaque(wo0);
#pragma unroll
			for (int g = 0; g < KPT / G; ++g) {
				if (g < (int)ng) {
#pragma unroll
					for (int r = g * G; r < (g + 1) * G; ++r) {
						const u32 i = wo + r * 64;
						if constexpr (K16)
							kr[r] = i < cnt ? (KT)(upper16 | (KT)kp16[i]) : (KT)~(KT)0;
						else
							kr[r] = i < cnt ? kp[i] : kdf_invert((KT)~(KT)0, ka);
						vr[r] = i < cnt ? vp[i] : (VT)0;
					}
				}
			}
		}
		u64 keep[KPT];   // derived key : payload (padding: all-ones keys, last in memory order: they stay behind the leaf's pairs)
#pragma unroll
		for (int r = 0; r < KPT; ++r)
			keep[r] = ((u64)(K16 ? kr[r] : kdf_apply(kr[r], ka)) << 32) | (u64)vr[r];
		for (u32 c = 0; c < nrem; ++c) {
			const u32 shift = 32 + 8 * ((colpack >> (4 * c)) & 15u);
#pragma unroll
			for (int k = 0; k < 4; ++k)
				cell[wid][lane + 64 * k] = 0;
			u32 *wc = cell[wid];
			u32 rk[KPT / 2];
#pragma unroll
			for (int i = 0; i < KPT / 2; ++i)
				rk[i] = 0;
#pragma unroll
			for (int g = 0; g < KPT / G; ++g) {
				if (g < (int)ng) {
#pragma unroll
					for (int r = g * G; r < (g + 1) * G; ++r) {
						const u32 old = __hip_atomic_fetch_add(&wc[(u32)(keep[r] >> shift) & 0xFFu], 1u, __ATOMIC_RELAXED,
						                                       __HIP_MEMORY_SCOPE_WORKGROUP);
						rk[r >> 1] |= old << (16 * (r & 1));
					}
				}
			}
			__syncthreads();
			u32 tot = 0, incl = 0;
			if (tid < 256) {
#pragma unroll
				for (int w = 0; w < NW; ++w)
					tot += cell[w][tid];
				u32 x = tot;
#pragma unroll
				for (int off = 1; off < 64; off <<= 1) {
					const u32 y = __shfl_up(x, off);
					if (lane >= (u32)off)
						x += y;
				}
				incl = x;
				if (lane == 63)
					wsum[wid] = x;
			}
			__syncthreads();
			if (tid < 256) {
				u32 acc = incl - tot;
				for (u32 w = 0; w < wid; ++w)
					acc += wsum[w];
#pragma unroll
				for (int w = 0; w < NW; ++w) {
					const u32 k = cell[w][tid];
					cell[w][tid] = acc;
					acc += k;
				}
			}
			__syncthreads();
#pragma unroll
			for (int g = 0; g < KPT / G; ++g) {
				if (g < (int)ng) {
#pragma unroll
					for (int r = g * G; r < (g + 1) * G; ++r) {
						const u32 pos = wc[(u32)(keep[r] >> shift) & 0xFFu] + ((rk[r >> 1] >> (16 * (r & 1))) & 0xFFFFu);
						stage[pos] = keep[r];
					}
				}
			}
			__syncthreads();
			if (c + 1 < nrem) {
				const u32 wo = opaque(wo0);
#pragma unroll
				for (int g = 0; g < KPT / G; ++g) {
					if (g < (int)ng) {
#pragma unroll
						for (int r = g * G; r < (g + 1) * G; ++r)
							keep[r] = stage[wo + r * 64];
					}
				}
			}
		}
		// write out: two pairs (16 bytes of LDS) per lane and step; payloads (and keys) to the dense result
		VT *vo = vout + ls.beg;
		KT *ko = kout ? kout + ls.beg : nullptr;
		for (u32 i0 = tid * 2; i0 < cnt; i0 += BLOCK * 2) {
			typedef u64 pvec_t __attribute__((ext_vector_type(2)));
			const pvec_t x = *(const pvec_t *)&stage[i0];
			vo[i0] = (VT)(u32)x[0];
			if (ko)
				ko[i0] = kdf_invert((KT)(x[0] >> 32), ka);
			if (i0 + 1 < cnt) {
				vo[i0 + 1] = (VT)(u32)x[1];
				if (ko)
					ko[i0 + 1] = kdf_invert((KT)(x[1] >> 32), ka);
			}
		}
		__syncthreads();   // the staged leaf has been read before the next one is staged
	}
}

}  // namespace rsx
