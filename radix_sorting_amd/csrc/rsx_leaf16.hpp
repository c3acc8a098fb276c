// rsx_leaf16.hpp -- the leaves of a keys-only two-level sort whose slots hold TWO-BYTE values (rsx_hybrid.hpp, DENSE), gfx950.
//
// What such a leaf has to do: a slot holds up to `slack_cap` 16-bit values -- the low half of the derived keys of one
// (digit, digit) bucket, in any order -- and the dense result wants them ascending, widened to the caller's element images
// (the upper half follows from the slot, kdf_invert gives the caller's bits back).  The reference's two remaining passes
// (radix_sort.hpp:82-90, columns 0 and 1) produce exactly that order; keys that compare equal are the same bits, so ANY
// ascending order is the reference's output, and nothing here has to be stable.
//
// rsx_leaf_sort_kernel does it the reference's way, two stable 8-bit passes through the LDS: per key and column a returning
// atomic, a read of the run start and a scattered store -- six data-dependent LDS operations per key, 59 % of the LDS cycles
// lost to bank conflicts, 0.33-0.36 of the HBM peak (profiles/r03/bench/roofline_table.json).  This kernel does it in ONE
// placement and two register passes:
//
//   1. place by the TOP TWELVE bits: one returning atomic on a cell per 12-bit bin shared by the whole workgroup (the order
//      among the keys of a bin does not matter, so there is no row per wave), a scan of the 4096 cells, one read of the bin's
//      start, one 2-byte store.  With ~4096 keys in 4096 bins a bin holds one key on average and every key then lies within
//      (its bin's size - 1) places of where it belongs.
//   2. finish the low four bits in registers: every lane sorts 16 consecutive staged values with a sorting network on packed
//      16-bit halves (v_pk_min_u16 / v_pk_max_u16: an 8-input network on both halves at once, then a bitonic merge of the two
//      halves: 98 instructions per 16 keys), then merges the upper half of its chunk with the lower half of the next lane's
//      (a wave shift; 68 instructions): chunks at 16 i, then at 16 i + 8.  A bin of at most 9 keys is cut by at most one
//      chunk boundary and lies inside the shifted chunk around that boundary, so after the two passes every bin is in order;
//      every further pair of passes (through the LDS; rare) takes bins of 16 more keys.  The scan knows the largest bin:
//      a leaf with a bin of more than MAXBIN2 keys does nothing here and is put on a list, which a launch of
//      rsx_leaf_sort_kernel works off afterwards (evenly spread keys: never; the list is there for keys that cluster in
//      their low sixteen bits although the sample let them pass).
//
// Three data-dependent LDS operations per key instead of six, and the rest is linear 16-byte traffic.
#pragma once

#include "rsx_hybrid.hpp"

namespace rsx {

typedef unsigned short u16x2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ u32 pk_min_u16(u32 a, u32 b)
{
	return __builtin_bit_cast(u32, __builtin_elementwise_min(__builtin_bit_cast(u16x2_t, a), __builtin_bit_cast(u16x2_t, b)));
}
__device__ __forceinline__ u32 pk_max_u16(u32 a, u32 b)
{
	return __builtin_bit_cast(u32, __builtin_elementwise_max(__builtin_bit_cast(u16x2_t, a), __builtin_bit_cast(u16x2_t, b)));
}
__device__ __forceinline__ u32 rot16(u32 x) { return __builtin_amdgcn_alignbit(x, x, 16); }
// (lo(a), lo(b)) and (hi(a), hi(b)) as one register each
__device__ __forceinline__ u32 lo_lo(u32 a, u32 b) { return __builtin_amdgcn_perm(b, a, 0x05040100u); }
__device__ __forceinline__ u32 hi_hi(u32 a, u32 b) { return __builtin_amdgcn_perm(b, a, 0x07060302u); }

// d[i] = (L_i, H_i) with L_0 <= ... <= L_7 in the low halves and H_0 <= ... <= H_7 in the high halves: afterwards the sixteen
// values ascending in memory order, d[i] = value 2i (low half) and value 2i + 1 (high half).  (L_i, H_{7-i}) is a bitonic
// sequence of 16; its merge needs one register rotated per pair of comparators down to distance 2 and a transposition for
// distance 1 (tools/ubench/leaf16_net.py checks the arrangement over all 2^16 zero-one inputs).  60 instructions.
__device__ __forceinline__ void merge16_packed(u32 (&d)[8])
{
	u32 q[2][4];   // q[0][i] = (X_i, X_{7-i}): the eight smaller values, bitonic; q[1][i] the eight larger ones
#pragma unroll
	for (int i = 0; i < 4; ++i) {
		const u32 s = rot16(d[7 - i]);
		q[0][i] = pk_min_u16(d[i], s);
		q[1][i] = pk_max_u16(d[i], s);
	}
#pragma unroll
	for (int h = 0; h < 2; ++h) {
		u32 quad[2][2];   // distance 4: quad[0] = the four smaller, as (p0, p3), (p1, p2); quad[1] the four larger
#pragma unroll
		for (int i = 0; i < 2; ++i) {
			const u32 s = rot16(q[h][3 - i]);
			quad[0][i] = pk_min_u16(q[h][i], s);
			quad[1][i] = pk_max_u16(q[h][i], s);
		}
#pragma unroll
		for (int g = 0; g < 2; ++g) {
			const u32 s = rot16(quad[g][1]);
			const u32 mn = pk_min_u16(quad[g][0], s), mx = pk_max_u16(quad[g][0], s);   // distance 2: places (0, 1) and (2, 3)
			const u32 t1 = lo_lo(mn, mx), t2 = hi_hi(mn, mx);                           // (0, 2) and (1, 3)
			const u32 m2 = pk_min_u16(t1, t2), x2 = pk_max_u16(t1, t2);                 // distance 1
			d[4 * h + 2 * g] = lo_lo(m2, x2);
			d[4 * h + 2 * g + 1] = hi_hi(m2, x2);
		}
	}
}

// Sixteen 16-bit values in eight registers, memory order as above, in any order: ascending afterwards.  Batcher's
// 19-comparator network on the registers sorts the low halves and the high halves (8 values each) at once, then the merge.
__device__ __forceinline__ void sort16_packed(u32 (&d)[8])
{
#define RSX_CE(i, j)                           \
	{                                          \
		const u32 t_ = pk_min_u16(d[i], d[j]); \
		d[j] = pk_max_u16(d[i], d[j]);         \
		d[i] = t_;                             \
	}
	RSX_CE(0, 1) RSX_CE(2, 3) RSX_CE(4, 5) RSX_CE(6, 7)
	RSX_CE(0, 2) RSX_CE(1, 3) RSX_CE(4, 6) RSX_CE(5, 7)
	RSX_CE(1, 2) RSX_CE(5, 6)
	RSX_CE(0, 4) RSX_CE(1, 5) RSX_CE(2, 6) RSX_CE(3, 7)
	RSX_CE(2, 4) RSX_CE(3, 5)
	RSX_CE(1, 2) RSX_CE(3, 4) RSX_CE(5, 6)
#undef RSX_CE
	merge16_packed(d);
}

// a[0..3] and b[0..3]: eight ascending values each, memory order -> d[0..7] as merge16_packed wants them
__device__ __forceinline__ void planes_of_two_runs(u32 (&d)[8], const u32 (&a)[4], const u32 (&b)[4])
{
#pragma unroll
	for (int m = 0; m < 4; ++m) {
		d[2 * m] = lo_lo(a[m], b[m]);
		d[2 * m + 1] = hi_hi(a[m], b[m]);
	}
}

// inclusive prefix sum over the 64 lanes of a wave (DPP: rows of 16 by shifts, then the rows' totals broadcast)
__device__ __forceinline__ u32 wave_incl_scan_dpp(u32 x)
{
	x += (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x111 /* row_shr:1 */, 0xF, 0xF, true);
	x += (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x112 /* row_shr:2 */, 0xF, 0xF, true);
	x += (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x114 /* row_shr:4 */, 0xF, 0xF, true);
	x += (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x118 /* row_shr:8 */, 0xF, 0xF, true);
	x += (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x142 /* row_bcast:15 */, 0xA, 0xF, false);
	x += (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x143 /* row_bcast:31 */, 0xC, 0xF, false);
	return x;
}
// lane l gets lane l + 1's value (lane 63: unspecified)
__device__ __forceinline__ u32 from_next_lane(u32 x)
{
	return (u32)__builtin_amdgcn_update_dpp((int)x, (int)x, 0x130 /* wave_shl:1 */, 0xF, 0xF, false);
}

// Every kernel of this file is launched with ONE leaf per workgroup (per wave, per row of sixteen lanes): the grid covers the leaf
// table (65536 entries, one per level-2 slot).  The loops over the table stay, for a smaller grid, but end after their first
// trip: written as open loops, everything a leaf computes from thread indices and kernel arguments is loop-invariant, the
// compiler hoists it in front of a loop that runs once and then spills it -- rsx_leafk_kernel<u64, u64> carried 20 bytes of
// scratch and two spilled SGPRs per lane that way, 80 registers instead of 50 (round 5: tools/ubench/leafk_probe, 1.47 ->
// 1.29 ms for the leaves of 2^28 u64 keys from this alone).  false: the open loops (a grid smaller than the table works again).
constexpr bool LEAF_ONE_PER_GROUP = true;

// the last places of a two-byte slot that rsx_pass16a_kernel (rsx_pass16.hpp) fills from both ends
constexpr u32 LEAF16_BACK = 128;

template <int BLOCK_, int CAP_, int WPE_, int NBITS_ = 12, int SKIP_ = 0> struct Leaf16Cfg {
	static constexpr int BLOCK = BLOCK_, CAP = CAP_, WPE = WPE_, NW = BLOCK_ / 64;
	static constexpr int SKIP = SKIP_;   // probe only: 1 no register passes, 2 no count / scan / placement, 4 no write-out
	static constexpr int NV = (CAP / 8 + BLOCK - 1) / BLOCK;     // 16-byte vectors of eight values per lane
	static constexpr int NCH = (CAP / 16 + BLOCK - 1) / BLOCK;   // chunks of sixteen values per lane
	static constexpr int NBITS = NBITS_;                         // the top NBITS of the sixteen bits name a value's bin:
	static constexpr int NBIN = 1 << NBITS;                      // about as many bins as the leaf has keys
	static constexpr int NCELLW = NBIN / 2;                      // two 16-bit cells to a word
	static constexpr int PLANES = NCELLW / 4 / BLOCK;            // 16-byte vectors of cells per thread
	static constexpr u32 MAXBIN = 9;     // the largest bin two passes over 16-value chunks put right
	static constexpr u32 MAXBIN2 = 25;   // ... and four passes (ceil((m - 1) / 8) + 1 passes for a bin of m keys)
	// who takes the leaves this kernel leaves alone: rsx_leaf_sort_kernel (byte columns only: with MSB digits at other bit positions
	// this kernel has to go on until the leaf is in order) or, behind the shapes for more than 5120 values, rsx_leafc_kernel (any digits)
	static constexpr bool REDO_ANY_SHIFT = CAP > 5120;
	static_assert(NBITS >= 10 && NBITS <= 14, "");
	static_assert(CAP % 16 == 0 && CAP <= 40960, "whole chunks; bin starts fit 16 bits");
	static_assert(PLANES == 1 || PLANES == 2, "one or two vectors of cells per thread");
	static_assert(NCELLW == 4 * BLOCK * PLANES, "every cell in some thread's vectors");
};

// segtab[s].slot names the slot (of slack_cap two-byte values) leaf s reads; the sorted keys go to out + segtab[s].beg.
// Does nothing unless the device-side plan is a two-level one whose leaves lie in slots of (lo, hi] values.
// redo / SegCtl::nredo: the leaves this kernel leaves alone (a bin of more than maxbin2 <= MAXBIN2 keys; 0: every leaf, tests).
// SegCtl::leaf16 == 0 (the sample saw the keys cluster in their low sixteen bits): nothing at all.
template <typename KT, typename C>
__global__ __launch_bounds__(C::BLOCK, C::WPE) void rsx_leaf16_kernel(KT *__restrict__ src, KT *__restrict__ aux,
                                                                      const Plan *__restrict__ plan,
                                                                      const LeafSeg *__restrict__ segtab, SegCtl *__restrict__ ctl,
                                                                      KdfArgs<KT> ka, u32 lo, u32 hi,
                                                                      const uint16_t *__restrict__ slots, u32 slack_cap,
                                                                      u32 *__restrict__ redo, u32 maxbin2 = C::MAXBIN2)
{
	static_assert(sizeof(KT) == 4, "4-byte keys: two MSB digits in the slot, two bytes in the leaf");
	constexpr int BLOCK = C::BLOCK, CAP = C::CAP, NV = C::NV, NCH = C::NCH, NCELLW = C::NCELLW, NW = C::NW, PLANES = C::PLANES;
	const u32 hyb = plan->hyb, ncols = plan->ncols;
	const u32 sh1 = ctl->shift1, sh2 = ctl->shift2;   // the MSB digits' bit positions (24 and 16 unless the keys' top bits are constant)
	const u32 mode = ctl->mode, maxleaf = ctl->maxleaf, nseg = ctl->nleaf, on = ctl->leaf16;
	if (hyb != HYB_TWO_LEVEL || ncols != 4 || mode != SEG_MODE_LEAVES || maxleaf <= lo || maxleaf > hi || !on)
		return;
	KT *out = src;   // (four kept columns: the reference's passes end in src, radix_sort.hpp:92)
	(void)aux;
	// cells: 16 bits per bin, bin b in half (b & 1) of word b >> 1: a count, then the bin's start, then its cursor
	__shared__ __attribute__((aligned(16))) u32 cell[NCELLW + 64];           // + a word per lane for values that do not exist
	__shared__ __attribute__((aligned(16))) uint16_t stage[CAP + 32 + 64];   // + padding behind the leaf + a place per lane
	__shared__ u32 ws[NW], wmax[NW];
	const u32 tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
	const u32 swid = (u32)__builtin_amdgcn_readfirstlane((int)wid);
	const KT key0 = (KT)ctl->key0_lo;
	// the leaf's values: the sh2 bits below the MSB digits (sixteen of them unless ...); their bins: the top NBITS of those
	const u32 D = sh2 > (u32)C::NBITS ? sh2 - (u32)C::NBITS : 0u, nb = sh2 - D;
	const KT above = sh1 + 8 >= 32u ? (KT)0 : (KT)(key0 >> (sh1 + 8) << (sh1 + 8));   // what every key has above the level-1 digit
	for (u32 s = blockIdx.x; s < nseg; s += gridDim.x) {
		[&]() {   // (one leaf; see LEAF_ONE_PER_GROUP)
		const LeafSeg ls = segtab[s];
		const u32 cnt = ls.cnt, slot = ls.slot;
		if (cnt == 0)
			return;   // (next leaf)
		// A slot filled by rsx_pass16a_kernel (rsx_pass16.hpp) holds its values at BOTH ends: `front` values from its beginning on
		// (whole 64-byte atoms) and `back` (LeafSeg::ncols >> 16, at most LEAF16_BACK) in its last LEAF16_BACK places -- what
		// that pass still carried when a workgroup's range of tiles ended.  back == 0: a slot as every other pass fills it.
		const u32 back = ls.ncols >> 16, front = cnt - back;
		// ---- the slot's values: 16 bytes per lane and step, all requested at once
		const uint16_t *q = slots + (u64)(slot - 1) * slack_cap;
		// (the vectors of the back follow the front's: vector v of the leaf is the front's for v < VF, the back's vector v - VF behind)
		const u32 VF = (front + 7u) >> 3, VB = (back + 7u) >> 3;
		u32x4 kv[NV];
		int nvalid[NV];
#pragma unroll
		for (int j = 0; j < NV; ++j) {
			const u32 v = tid + BLOCK * j;
			const bool isback = v >= VF;
			const u32 e0 = 8 * (isback ? v - VF : v);
			const int left = (int)(isback ? back : front) - (int)e0;
			nvalid[j] = left < 0 ? 0 : left > 8 ? 8 : left;
			kv[j] = u32x4{0, 0, 0, 0};
			if (left > 0)
				kv[j] = *(const u32x4 *)(q + (isback ? slack_cap - LEAF16_BACK : 0u) + e0);
		}
		// vectors in which this WAVE has any value (the last round of a slot that is not full)
		auto wave_has = [&](int j) { return 64 * swid + BLOCK * (u32)j < VF + VB; };
		{
			const u32x4 zero = {0, 0, 0, 0};
#pragma unroll
			for (int j = 0; j < PLANES; ++j)
				((u32x4 *)cell)[tid + BLOCK * j] = zero;
		}
		__syncthreads();
		u32 mx = 0;
		if constexpr (C::SKIP & 2) {
#pragma unroll
			for (int j = 0; j < NV; ++j)
				if (wave_has(j))
					*(u32x4 *)&stage[8 * (tid + BLOCK * j)] = kv[j];
		} else {
			// a value's cell: byte address of its word and the shift of its half; values that do not exist count in the lane's own word
			auto cell_of = [&](u32 w, int k, bool valid, u32 &sh) -> u32 * {
				// k even: the value is w[15:0], k odd: w[31:16]; its bin: nb bits from bit D of the value on (one bit-field extract)
				const u32 o = D + 16u * (u32)(k & 1);   // (uniform)
				sh = __builtin_amdgcn_ubfe(w, o, 1u) << 4;
				return &cell[valid ? __builtin_amdgcn_ubfe(w, o + 1u, nb - 1u) : NCELLW + lane];
			};
			// ---- count
#pragma unroll
			for (int j = 0; j < NV; ++j) {
				if (wave_has(j)) {
#pragma unroll
					for (int k = 0; k < 8; ++k) {
						u32 sh;
						u32 *a = cell_of(kv[j][k >> 1], k, k < nvalid[j], sh);
						__hip_atomic_fetch_add(a, 1u << sh, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
					}
				}
			}
			__syncthreads();
			// ---- scan: thread t owns the 16-byte vectors t, t + BLOCK, ... of the cells (conflict-free); the order of the bins
			// is vector-major, so one scan over the threads per vector is needed: 16-bit sums packed into one register
			u32x4 c[PLANES];
			u32 pk = 0, mxp = 0;
#pragma unroll
			for (int j = 0; j < PLANES; ++j) {
				c[j] = ((const u32x4 *)cell)[tid + BLOCK * j];
				u32 run = 0;
#pragma unroll
				for (int i = 0; i < 4; ++i) {
					const u32 x = c[j][i];
					mxp = pk_max_u16(mxp, x);
					const u32 lo16 = x & 0xFFFFu, hs = run + lo16;
					c[j][i] = run | (hs << 16);   // the two bins' starts, relative to the vector's
					run = hs + (x >> 16);
				}
				pk |= run << (16 * j);
			}
			mx = (mxp & 0xFFFFu) > (mxp >> 16) ? (mxp & 0xFFFFu) : (mxp >> 16);
			const u32 incl = wave_incl_scan_dpp(pk);
			// the workgroup's largest bin
#pragma unroll
			for (int o = 32; o > 0; o >>= 1) {
				const u32 y = (u32)__shfl_xor((int)mx, o);
				mx = mx > y ? mx : y;
			}
			if (lane == 63) {
				ws[wid] = incl;
				wmax[wid] = mx;
			}
			__syncthreads();
			mx = wmax[0];
#pragma unroll
			for (int w = 1; w < NW; ++w)
				mx = mx > wmax[w] ? mx : wmax[w];
			if (mx > maxbin2 && (C::REDO_ANY_SHIFT || !(sh1 & 7u))) {
				// a bin too large for the register passes: the leaf goes to rsx_leaf_sort_kernel (nothing was written) -- which
				// sorts by byte columns: with MSB digits at other bit positions this kernel goes on until the leaf is in order
				if (tid == 0)
					redo[atomicAdd(&ctl->nredo, 1u)] = s;
				// (rsx_leaf_sort_kernel reads dense slots: the back's values move up behind the front's -- they are all in registers)
				if (back) {
#pragma unroll
					for (int j = 0; j < NV; ++j) {
						const u32 v = tid + BLOCK * j;
						if (v >= VF) {
							uint16_t *qw = const_cast<uint16_t *>(q) + front + 8 * (v - VF);
#pragma unroll
							for (int k2 = 0; k2 < 8; ++k2)
								if (k2 < nvalid[j])
									qw[k2] = (uint16_t)(kv[j][k2 >> 1] >> (16 * (k2 & 1)));
						}
					}
				}
				return;   // (next leaf)
			}
			{
				u32 base = 0, tot = 0;
#pragma unroll
				for (u32 w = 0; w < (u32)NW; ++w) {
					const u32 a = ws[w];
					base += w < wid ? a : 0u;
					tot += a;
				}
				const u32 e = incl - pk + base;   // exclusive, per 16-bit field
				u32 o[2];
				o[0] = e & 0xFFFFu;
				o[1] = (tot & 0xFFFFu) + (e >> 16);
#pragma unroll
				for (int j = 0; j < PLANES; ++j) {
					const u32 bb = o[j] | (o[j] << 16);
					u32x4 x;
#pragma unroll
					for (int i = 0; i < 4; ++i)
						x[i] = c[j][i] + bb;
					((u32x4 *)cell)[tid + BLOCK * j] = x;
				}
			}
			__syncthreads();
			// ---- place: the returning atomic on the bin's start is the key's place (any order inside a bin)
#pragma unroll
			for (int j = 0; j < NV; ++j) {
				if (wave_has(j)) {
#pragma unroll
					for (int k = 0; k < 8; ++k) {
						const u32 w = kv[j][k >> 1];
						const bool valid = k < nvalid[j];
						u32 sh;
						u32 *a = cell_of(w, k, valid, sh);
						const u32 old = __hip_atomic_fetch_add(a, 1u << sh, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
						const u32 pos = (old >> sh) & 0xFFFFu;
						stage[valid ? pos : CAP + 32 + lane] = (uint16_t)((k & 1) ? (w >> 16) : w);
					}
				}
			}
		}
		if (tid < 32)
			stage[cnt + tid] = (uint16_t)0xFFFFu;   // what the last chunks read behind the leaf's end sorts last
		__syncthreads();
		if constexpr (!(C::SKIP & 1)) {
			// ---- the low four bits: chunks of 16 values at 16 i (sorted), then at 16 i + 8 (two sorted halves: merged), and
			// twice more for bins of more than MAXBIN keys (evenly spread keys: one leaf in two thousand)
			// ceil((m - 1) / 8) + 1 passes for a bin of m: the blocks of eight it can span (2 up to 9 values, 3 up to 17, 4 up to 25)
			const u32 npass = mx <= C::MAXBIN ? 2u : mx <= 17u ? 3u : mx <= C::MAXBIN2 ? 4u : 2u * (((mx + 6) / 8 + 2) / 2);
			for (u32 pass = 0; pass < npass; ++pass) {
				const u32 off = 8 * (pass & 1);
#pragma unroll
				for (int r = 0; r < NCH; ++r) {
					const u32 ch = tid + BLOCK * r;
					if (16 * ch + off < cnt) {
						u32x4 *p = (u32x4 *)&stage[16 * ch + off];
						const u32x4 x0 = p[0], x1 = p[1];
						u32 d[8];
						if (pass == 0) {
							d[0] = x0[0], d[1] = x0[1], d[2] = x0[2], d[3] = x0[3];
							d[4] = x1[0], d[5] = x1[1], d[6] = x1[2], d[7] = x1[3];
							sort16_packed(d);
						} else {
							const u32 a[4] = {x0[0], x0[1], x0[2], x0[3]}, b[4] = {x1[0], x1[1], x1[2], x1[3]};
							planes_of_two_runs(d, a, b);   // (both halves were sorted by the pass before)
							merge16_packed(d);
						}
						p[0] = u32x4{d[0], d[1], d[2], d[3]};
						p[1] = u32x4{d[4], d[5], d[6], d[7]};
					}
				}
				__syncthreads();
			}
		}
		// ---- write out: four values (8 bytes of LDS) -> four keys (16 bytes) per lane and step
		if constexpr (!(C::SKIP & 4)) {
			const KT upper = (KT)(above | ((KT)((slot - 1) >> 8) << sh1) | ((KT)((slot - 1) & 255u) << sh2));
			KT *o = out + ls.beg;
			for (u32 i0 = 4 * tid; i0 < cnt; i0 += 4 * BLOCK) {
				const uint2 x = *(const uint2 *)&stage[i0];
				KT kk[4];
				kk[0] = kdf_invert((KT)(upper | __builtin_amdgcn_ubfe(x.x, 0u, sh2)), ka);
				kk[1] = kdf_invert((KT)(upper | __builtin_amdgcn_ubfe(x.x, 16u, sh2)), ka);
				kk[2] = kdf_invert((KT)(upper | __builtin_amdgcn_ubfe(x.y, 0u, sh2)), ka);
				kk[3] = kdf_invert((KT)(upper | __builtin_amdgcn_ubfe(x.y, 16u, sh2)), ka);
				if (i0 + 4 <= cnt) {
					store_chunk<KT, 4>(o + i0, kk);
				} else {
#pragma unroll
					for (int e = 0; e < 4; ++e)
						if (i0 + e < cnt)
							o[i0 + e] = kk[e];
				}
			}
		}
		// (the next leaf's first writes to `stage` lie behind three more barriers)
		}();
		if constexpr (LEAF_ONE_PER_GROUP)
			break;
	}
}

// ---- small slots: a WAVE per leaf ---------------------------------------------------------------------------------------
// Arrays of 8 Mi .. 50 Mi keys (the reference's own headline, 4 * 10^7 keys, radix_bench.cpp:135-138, among them) leave slots of
// a few hundred values: a workgroup per leaf then spends its time in barriers and in being launched (65536 leaves of 610
// values: 148 us, 1.6 TB/s).  Here a leaf is one wave's: the same algorithm, everything in the wave's own part of the LDS --
// a wave's DS operations execute in order, so nothing waits for anybody -- the scan is one DPP scan, the second register pass
// takes the next chunk's lower half from the next LANE, and a workgroup's waves work on leaves of their own.
template <int CAP_, int NBITS_, int WAVES_> struct Leaf16WCfg {
	static constexpr int CAP = CAP_, NBITS = NBITS_, NW = WAVES_, BLOCK = 64 * WAVES_;
	static constexpr int NV = CAP / 512;                 // 16-byte vectors of eight values per lane
	static constexpr int NBIN = 1 << NBITS, NCELLW = NBIN / 2;
	static constexpr int PLANES = NCELLW / 4 / 64;       // 16-byte vectors of cells per lane
	static constexpr int NCL = CAP > 1024 ? CAP / 1024 : 1;   // chunks of sixteen values per lane in the register passes
	static constexpr u32 MAXBIN = 9;
	static_assert(CAP == 512 || CAP == 1024 || CAP == 2048, "one or two chunks of sixteen values per lane");
	static_assert(PLANES == 1 || PLANES == 2, "");
};

template <typename KT, typename C>
__global__ __launch_bounds__(C::BLOCK, 8) void rsx_leaf16w_kernel(KT *__restrict__ src, KT *__restrict__ aux,
                                                                  const Plan *__restrict__ plan,
                                                                  const LeafSeg *__restrict__ segtab, SegCtl *__restrict__ ctl,
                                                                  KdfArgs<KT> ka, u32 lo, u32 hi,
                                                                  const uint16_t *__restrict__ slots, u32 slack_cap)
{
	static_assert(sizeof(KT) == 4, "4-byte keys: two MSB digits in the slot, two bytes in the leaf");
	constexpr int CAP = C::CAP, NV = C::NV, NCELLW = C::NCELLW, NW = C::NW, PLANES = C::PLANES;
	const u32 hyb = plan->hyb, ncols = plan->ncols;
	const u32 sh1 = ctl->shift1, sh2 = ctl->shift2;   // the MSB digits' bit positions (rsx_leaf16_kernel)
	const u32 mode = ctl->mode, maxleaf = ctl->maxleaf, nseg = ctl->nleaf;
	// (takes every leaf, whatever the sample made of the low sixteen bits: SegCtl::leaf16 is for the workgroup kernel)
	if (hyb != HYB_TWO_LEVEL || ncols != 4 || mode != SEG_MODE_LEAVES || maxleaf <= lo || maxleaf > hi)
		return;
	KT *out = src;   // (four kept columns: the reference's passes end in src, radix_sort.hpp:92)
	(void)aux;
	__shared__ __attribute__((aligned(16))) u32 cell_all[NW][NCELLW + 64];
	__shared__ __attribute__((aligned(16))) uint16_t stage_all[NW][CAP + 32 + 64];
	const u32 lane = threadIdx.x & 63;
	const u32 swid = (u32)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
	u32 *cell = cell_all[swid];
	uint16_t *stage = stage_all[swid];
	const KT key0 = (KT)ctl->key0_lo;
	const u32 D = sh2 > (u32)C::NBITS ? sh2 - (u32)C::NBITS : 0u, nb = sh2 - D;
	const KT above = sh1 + 8 >= 32u ? (KT)0 : (KT)(key0 >> (sh1 + 8) << (sh1 + 8));
	for (u32 s = blockIdx.x * NW + swid; s < nseg; s += gridDim.x * NW) {
		[&]() {   // (one leaf; see LEAF_ONE_PER_GROUP)
		const LeafSeg ls = segtab[s];
		const u32 cnt = ls.cnt, slot = ls.slot;
		if (cnt == 0)
			return;   // (next leaf)
		const uint16_t *q = slots + (u64)(slot - 1) * slack_cap;
		// (a slot filled by rsx_pass16a_kernel holds its values at both ends, as rsx_leaf16_kernel reads them: the front's vectors,
		// then the back's -- slots of more than 1024 values; back == 0 otherwise)
		const u32 back = ls.ncols >> 16, front = cnt - back;
		const u32 VF = (front + 7u) >> 3, VB = (back + 7u) >> 3;
		u32x4 kv[NV];
		int nvalid[NV];
#pragma unroll
		for (int j = 0; j < NV; ++j) {
			const u32 v = lane + 64 * j;
			const bool isback = v >= VF;
			const u32 e0 = 8 * (isback ? v - VF : v);
			const int left = (int)(isback ? back : front) - (int)e0;
			nvalid[j] = left < 0 ? 0 : left > 8 ? 8 : left;
			kv[j] = u32x4{0, 0, 0, 0};
			if (left > 0)
				kv[j] = *(const u32x4 *)(q + (isback ? slack_cap - LEAF16_BACK : 0u) + e0);
		}
		auto wave_has = [&](int j) { return 64 * (u32)j < VF + VB; };
		{
			const u32x4 zero = {0, 0, 0, 0};
#pragma unroll
			for (int j = 0; j < PLANES; ++j)
				((u32x4 *)cell)[lane + 64 * j] = zero;
		}
		RSX_COMPILER_FENCE();
		auto cell_of = [&](u32 w, int k, bool valid, u32 &sh) -> u32 * {
			const u32 o = D + 16u * (u32)(k & 1);
			sh = __builtin_amdgcn_ubfe(w, o, 1u) << 4;
			return &cell[valid ? __builtin_amdgcn_ubfe(w, o + 1u, nb - 1u) : NCELLW + lane];
		};
#pragma unroll
		for (int j = 0; j < NV; ++j) {
			if (wave_has(j)) {
#pragma unroll
				for (int k = 0; k < 8; ++k) {
					u32 sh;
					u32 *a = cell_of(kv[j][k >> 1], k, k < nvalid[j], sh);
					__hip_atomic_fetch_add(a, 1u << sh, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
				}
			}
		}
		RSX_COMPILER_FENCE();
		u32x4 c[PLANES];
		u32 pk = 0, mxp = 0;
#pragma unroll
		for (int j = 0; j < PLANES; ++j) {
			c[j] = ((const u32x4 *)cell)[lane + 64 * j];
			u32 run = 0;
#pragma unroll
			for (int i = 0; i < 4; ++i) {
				const u32 x = c[j][i];
				mxp = pk_max_u16(mxp, x);
				const u32 lo16 = x & 0xFFFFu, hs = run + lo16;
				c[j][i] = run | (hs << 16);
				run = hs + (x >> 16);
			}
			pk |= run << (16 * j);
		}
		u32 mx = (mxp & 0xFFFFu) > (mxp >> 16) ? (mxp & 0xFFFFu) : (mxp >> 16);
#pragma unroll
		for (int o = 32; o > 0; o >>= 1) {
			const u32 y = (u32)__shfl_xor((int)mx, o);
			mx = mx > y ? mx : y;
		}
		mx = (u32)__builtin_amdgcn_readfirstlane((int)mx);
		{
			const u32 incl = wave_incl_scan_dpp(pk);
			const u32 tot = (u32)__builtin_amdgcn_readlane((int)incl, 63);
			const u32 e = incl - pk;
			const u32 o[2] = {e & 0xFFFFu, (tot & 0xFFFFu) + (e >> 16)};
#pragma unroll
			for (int j = 0; j < PLANES; ++j) {
				const u32 bb = o[j] | (o[j] << 16);
				u32x4 x;
#pragma unroll
				for (int i = 0; i < 4; ++i)
					x[i] = c[j][i] + bb;
				((u32x4 *)cell)[lane + 64 * j] = x;
			}
		}
		RSX_COMPILER_FENCE();
#pragma unroll
		for (int j = 0; j < NV; ++j) {
			if (wave_has(j)) {
#pragma unroll
				for (int k = 0; k < 8; ++k) {
					const u32 w = kv[j][k >> 1];
					const bool valid = k < nvalid[j];
					u32 sh;
					u32 *a = cell_of(w, k, valid, sh);
					const u32 old = __hip_atomic_fetch_add(a, 1u << sh, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
					const u32 pos = (old >> sh) & 0xFFFFu;
					stage[valid ? pos : CAP + 32 + lane] = (uint16_t)((k & 1) ? (w >> 16) : w);
				}
			}
		}
		if (lane < 32)
			stage[cnt + lane] = (uint16_t)0xFFFFu;
		RSX_COMPILER_FENCE();
		// chunks of 16 values at 16 i (sorted), then at 16 i + 8 (the upper half merged with the next lane's lower half): a
		// bin of m keys is in order after ceil((m - 1) / 8) + 1 such passes (odd-even transposition over the half chunks it
		// touches) -- one round for bins of up to 9 keys, which is what evenly spread keys give; no leaf is handed on
		const u32 nch = (cnt + 15) >> 4;
		const u32 rounds = mx <= C::MAXBIN ? 1u : ((mx + 6) / 8 + 2) / 2;
		constexpr int NCL = C::NCL;   // (lane l holds chunk l -- and, slots of 2048 values, chunk 64 + l: the chunk after lane 63's is lane 0's second)
		for (u32 r = 0; r < rounds; ++r) {
			u32 d[NCL][8];
#pragma unroll
			for (int cl = 0; cl < NCL; ++cl) {
				const u32 ci = lane + 64u * (u32)cl;
				const u32x4 *p = (const u32x4 *)&stage[16 * ci];
				const u32x4 ones = {~0u, ~0u, ~0u, ~0u};
				const u32x4 x0 = ci < nch ? p[0] : ones, x1 = ci < nch ? p[1] : ones;
				d[cl][0] = x0[0], d[cl][1] = x0[1], d[cl][2] = x0[2], d[cl][3] = x0[3];
				d[cl][4] = x1[0], d[cl][5] = x1[1], d[cl][6] = x1[2], d[cl][7] = x1[3];
				sort16_packed(d[cl]);
			}
			if (lane == 0)
				*(u32x4 *)&stage[0] = u32x4{d[0][0], d[0][1], d[0][2], d[0][3]};   // (the first eight values are in place)
#pragma unroll
			for (int cl = 0; cl < NCL; ++cl) {
				const u32 ci = lane + 64u * (u32)cl;
				u32 lowa[4], nxt[4], e[8];
#pragma unroll
				for (int m = 0; m < 4; ++m) {
					lowa[m] = d[cl][4 + m];
					const u32 y = from_next_lane(d[cl][m]);
					// (the chunk after lane 63's: lane 0's next one -- still as sorted above --, or nothing)
					const u32 wrap = cl + 1 < NCL ? (u32)__builtin_amdgcn_readfirstlane((int)d[cl + 1 < NCL ? cl + 1 : cl][m]) : ~0u;
					nxt[m] = lane == 63 ? wrap : y;
				}
				planes_of_two_runs(e, lowa, nxt);
				merge16_packed(e);
				if (ci < nch) {
					u32x4 *p = (u32x4 *)&stage[16 * ci + 8];
					p[0] = u32x4{e[0], e[1], e[2], e[3]};
					p[1] = u32x4{e[4], e[5], e[6], e[7]};
				}
			}
			RSX_COMPILER_FENCE();
		}
		{
			const KT upper = (KT)(above | ((KT)((slot - 1) >> 8) << sh1) | ((KT)((slot - 1) & 255u) << sh2));
			KT *o = out + ls.beg;
			for (u32 i0 = 4 * lane; i0 < cnt; i0 += 4 * 64) {
				const uint2 x = *(const uint2 *)&stage[i0];
				KT kk[4];
				kk[0] = kdf_invert((KT)(upper | __builtin_amdgcn_ubfe(x.x, 0u, sh2)), ka);
				kk[1] = kdf_invert((KT)(upper | __builtin_amdgcn_ubfe(x.x, 16u, sh2)), ka);
				kk[2] = kdf_invert((KT)(upper | __builtin_amdgcn_ubfe(x.y, 0u, sh2)), ka);
				kk[3] = kdf_invert((KT)(upper | __builtin_amdgcn_ubfe(x.y, 16u, sh2)), ka);
				if (i0 + 4 <= cnt) {
					store_chunk<KT, 4>(o + i0, kk);
				} else {
#pragma unroll
					for (int e = 0; e < 4; ++e)
						if (i0 + e < cnt)
							o[i0 + e] = kk[e];
				}
			}
		}
		RSX_COMPILER_FENCE();
		}();
		if constexpr (LEAF_ONE_PER_GROUP)
			break;
	}
}

// ---- a ROW of sixteen lanes per leaf: slots of up to 256 values (arrays of 9 .. 13 Mi keys; radix_bench's 10^7) ---------------
// rsx_leaf16w_kernel gives a leaf of ~150 values a whole wave, of which nineteen lanes load and ten sort.  Here a wave takes
// four leaves, one per DPP row: sixteen values per lane (two 16-byte loads), 128 bins (four words of cells per lane), the scan
// and the maximum over the row (row_shr steps, no row_bcast), the exchange with the next lane masked at the row's end.  Rows do
// not wait for each other except in the number of rounds (the wave's maximum: a finished row's rounds change nothing).
template <int WAVES_> struct Leaf16QCfg {
	static constexpr int CAP = 256, NBITS = 7, NW = WAVES_, BLOCK = 64 * WAVES_, ROWS = 4 * WAVES_;
	static constexpr int NCELLW = (1 << NBITS) / 2;      // 64 words of two cells: four per lane
	static constexpr int CELL_ROW = NCELLW + 16;         // + a word per lane for values that do not exist
	static constexpr int STAGE_ROW = CAP + 32 + 16;      // + what the last chunk reads behind the leaf + the same
	static constexpr u32 MAXBIN = 9;
};

__device__ __forceinline__ u32 row_incl_scan_dpp(u32 x)
{
	x += (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x111 /* row_shr:1 */, 0xF, 0xF, true);
	x += (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x112 /* row_shr:2 */, 0xF, 0xF, true);
	x += (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x114 /* row_shr:4 */, 0xF, 0xF, true);
	x += (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x118 /* row_shr:8 */, 0xF, 0xF, true);
	return x;
}

template <typename KT, typename C>
__global__ __launch_bounds__(C::BLOCK, 8) void rsx_leaf16q_kernel(KT *__restrict__ src, KT *__restrict__ aux,
                                                                  const Plan *__restrict__ plan,
                                                                  const LeafSeg *__restrict__ segtab, SegCtl *__restrict__ ctl,
                                                                  KdfArgs<KT> ka, u32 lo, u32 hi,
                                                                  const uint16_t *__restrict__ slots, u32 slack_cap)
{
	static_assert(sizeof(KT) == 4, "4-byte keys: two MSB digits in the slot, two bytes in the leaf");
	constexpr int CAP = C::CAP, NCELLW = C::NCELLW, NW = C::NW;
	const u32 hyb = plan->hyb, ncols = plan->ncols;
	const u32 sh1 = ctl->shift1, sh2 = ctl->shift2;
	const u32 mode = ctl->mode, maxleaf = ctl->maxleaf, nseg = ctl->nleaf;
	if (hyb != HYB_TWO_LEVEL || ncols != 4 || mode != SEG_MODE_LEAVES || maxleaf <= lo || maxleaf > hi)
		return;
	KT *out = src;   // (four kept columns: radix_sort.hpp:92)
	(void)aux;
	__shared__ __attribute__((aligned(16))) u32 cell_all[C::ROWS][C::CELL_ROW];
	__shared__ __attribute__((aligned(16))) uint16_t stage_all[C::ROWS][C::STAGE_ROW];
	const u32 lane = threadIdx.x & 63, l16 = lane & 15u, row = lane >> 4, wid = threadIdx.x >> 6;
	u32 *cell = cell_all[4 * wid + row];
	uint16_t *stage = stage_all[4 * wid + row];
	const KT key0 = (KT)ctl->key0_lo;
	const u32 D = sh2 > (u32)C::NBITS ? sh2 - (u32)C::NBITS : 0u, nb = sh2 - D;
	const KT above = sh1 + 8 >= 32u ? (KT)0 : (KT)(key0 >> (sh1 + 8) << (sh1 + 8));
	for (u32 s0 = (blockIdx.x * NW + wid) * 4; s0 < nseg; s0 += gridDim.x * NW * 4) {
		[&]() {   // (one leaf; see LEAF_ONE_PER_GROUP)
		const u32 s = s0 + row;
		u32 cnt = 0, slot = 1, beg = 0;
		if (s < nseg) {
			const LeafSeg ls = segtab[s];
			cnt = ls.cnt;
			slot = cnt ? ls.slot : 1u;
			beg = ls.beg;
		}
		const uint16_t *q = slots + (u64)(slot - 1) * slack_cap;
		u32x4 kv[2];
		int nvalid[2];
#pragma unroll
		for (int j = 0; j < 2; ++j) {
			const u32 e0 = 8 * (l16 + 16 * j);
			const int left = (int)cnt - (int)e0;
			nvalid[j] = left < 0 ? 0 : left > 8 ? 8 : left;
			kv[j] = u32x4{0, 0, 0, 0};
			if (left > 0)
				kv[j] = *(const u32x4 *)(q + e0);
		}
		((u32x4 *)cell)[l16] = u32x4{0, 0, 0, 0};
		RSX_COMPILER_FENCE();
		auto cell_of = [&](u32 w, int k, bool valid, u32 &sh) -> u32 * {
			const u32 o = D + 16u * (u32)(k & 1);
			sh = __builtin_amdgcn_ubfe(w, o, 1u) << 4;
			return &cell[valid ? __builtin_amdgcn_ubfe(w, o + 1u, nb - 1u) : NCELLW + l16];
		};
#pragma unroll
		for (int j = 0; j < 2; ++j) {
#pragma unroll
			for (int k = 0; k < 8; ++k) {
				u32 sh;
				u32 *a = cell_of(kv[j][k >> 1], k, k < nvalid[j], sh);
				__hip_atomic_fetch_add(a, 1u << sh, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
			}
		}
		RSX_COMPILER_FENCE();
		u32x4 c = ((const u32x4 *)cell)[l16];
		u32 run = 0, mxp = 0;
#pragma unroll
		for (int i = 0; i < 4; ++i) {
			const u32 x = c[i];
			mxp = pk_max_u16(mxp, x);
			const u32 lo16 = x & 0xFFFFu, hs = run + lo16;
			c[i] = run | (hs << 16);
			run = hs + (x >> 16);
		}
		u32 mx = (mxp & 0xFFFFu) > (mxp >> 16) ? (mxp & 0xFFFFu) : (mxp >> 16);
#pragma unroll
		for (int o = 8; o > 0; o >>= 1) {   // (the row's fullest bin)
			const u32 y = (u32)__shfl_xor((int)mx, o);
			mx = mx > y ? mx : y;
		}
		{
			const u32 e = row_incl_scan_dpp(run) - run, bb = e | (e << 16);
			u32x4 x;
#pragma unroll
			for (int i = 0; i < 4; ++i)
				x[i] = c[i] + bb;
			((u32x4 *)cell)[l16] = x;
		}
		RSX_COMPILER_FENCE();
#pragma unroll
		for (int j = 0; j < 2; ++j) {
#pragma unroll
			for (int k = 0; k < 8; ++k) {
				const u32 w = kv[j][k >> 1];
				const bool valid = k < nvalid[j];
				u32 sh;
				u32 *a = cell_of(w, k, valid, sh);
				const u32 old = __hip_atomic_fetch_add(a, 1u << sh, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
				const u32 pos = (old >> sh) & 0xFFFFu;
				stage[valid ? pos : CAP + 32 + l16] = (uint16_t)((k & 1) ? (w >> 16) : w);
			}
		}
		stage[cnt + l16] = (uint16_t)0xFFFFu;
		stage[cnt + 16 + l16] = (uint16_t)0xFFFFu;
		RSX_COMPILER_FENCE();
		// (the rounds of rsx_leaf16w_kernel; as many as the wave's neediest row wants)
		const u32 nch = (cnt + 15) >> 4;
		u32 rounds = mx <= C::MAXBIN ? 1u : ((mx + 6) / 8 + 2) / 2;
		{
			u32 y = (u32)__shfl_xor((int)rounds, 16);
			rounds = rounds > y ? rounds : y;
			y = (u32)__shfl_xor((int)rounds, 32);
			rounds = rounds > y ? rounds : y;
			rounds = (u32)__builtin_amdgcn_readfirstlane((int)rounds);
		}
		for (u32 r = 0; r < rounds; ++r) {
			u32 d[8];
			{
				const u32x4 *p = (const u32x4 *)&stage[16 * l16];
				const u32x4 ones = {~0u, ~0u, ~0u, ~0u};
				const u32x4 x0 = l16 < nch ? p[0] : ones, x1 = l16 < nch ? p[1] : ones;
				d[0] = x0[0], d[1] = x0[1], d[2] = x0[2], d[3] = x0[3];
				d[4] = x1[0], d[5] = x1[1], d[6] = x1[2], d[7] = x1[3];
			}
			sort16_packed(d);
			if (l16 == 0)
				*(u32x4 *)&stage[0] = u32x4{d[0], d[1], d[2], d[3]};   // (the first eight values are in place)
			u32 lowa[4], nxt[4];
#pragma unroll
			for (int m = 0; m < 4; ++m) {
				lowa[m] = d[4 + m];
				const u32 y = from_next_lane(d[m]);
				nxt[m] = l16 == 15 ? ~0u : y;   // (the next lane is another leaf's)
			}
			planes_of_two_runs(d, lowa, nxt);
			merge16_packed(d);
			if (l16 < nch) {
				u32x4 *p = (u32x4 *)&stage[16 * l16 + 8];
				p[0] = u32x4{d[0], d[1], d[2], d[3]};
				p[1] = u32x4{d[4], d[5], d[6], d[7]};
			}
			RSX_COMPILER_FENCE();
		}
		{
			const KT upper = (KT)(above | ((KT)((slot - 1) >> 8) << sh1) | ((KT)((slot - 1) & 255u) << sh2));
			KT *o = out + beg;
			for (u32 i0 = 4 * l16; i0 < cnt; i0 += 4 * 16) {
				const uint2 x = *(const uint2 *)&stage[i0];
				KT kk[4];
				kk[0] = kdf_invert((KT)(upper | __builtin_amdgcn_ubfe(x.x, 0u, sh2)), ka);
				kk[1] = kdf_invert((KT)(upper | __builtin_amdgcn_ubfe(x.x, 16u, sh2)), ka);
				kk[2] = kdf_invert((KT)(upper | __builtin_amdgcn_ubfe(x.y, 0u, sh2)), ka);
				kk[3] = kdf_invert((KT)(upper | __builtin_amdgcn_ubfe(x.y, 16u, sh2)), ka);
				if (i0 + 4 <= cnt) {
					store_chunk<KT, 4>(o + i0, kk);
				} else {
#pragma unroll
					for (int e = 0; e < 4; ++e)
						if (i0 + e < cnt)
							o[i0 + e] = kk[e];
				}
			}
		}
		RSX_COMPILER_FENCE();
		}();
		if constexpr (LEAF_ONE_PER_GROUP)
			break;
	}
}

// ---- leaves of 8-byte keys (keys only): the same placement, the register passes on whole 4- or 8-byte values -----------------
// BASELINE.json's cfg 3: 2^28 u64 keys keep eight, five or four columns; two MSB passes leave a leaf six, three or two of
// them (48 / 24 / 16 bits), which rsx_leaf_sort_kernel goes through one LDS pass per column (six columns: the top three and
// odd-even transposition sweeps) at 0.28 of the HBM peak.  Here: the slot's keys (whole element images: the level-2 pass of
// 8-byte keys writes them unchanged) become derived keys cut to the carried type CT -- 4 bytes if the leaf's columns all lie
// in the low word, else 8 --, are placed by twelve bits (the leaf's highest column and the top nibble of the next one) and
// finished by Batcher's odd-even merge sort on sixteen values per lane: 63 comparators on aligned chunks, then the 25 of its
// last merge step on chunks shifted by 8, min / max for 4-byte values, compare + select for 8-byte ones.
template <int P, typename T, typename F> __device__ __forceinline__ void batcher_stage(T (&d)[16], F &&ce)
{
#pragma unroll
	for (int k = P; k >= 1; k /= 2) {
#pragma unroll
		for (int j = k % P; j <= 15 - k; j += 2 * k) {
#pragma unroll
			for (int i = 0; i <= (k - 1 < 15 - j - k ? k - 1 : 15 - j - k); ++i) {
				if ((i + j) / (2 * P) == (i + j + k) / (2 * P))
					ce(d[i + j], d[i + j + k]);
			}
		}
	}
}
template <typename T> __device__ __forceinline__ void ce_minmax(T &a, T &b)
{
	const T lo = a < b ? a : b, hi = a < b ? b : a;
	a = lo;
	b = hi;
}
// sixteen values in any order -> ascending; two ascending runs of eight -> ascending
template <typename T> __device__ __forceinline__ void sort16_values(T (&d)[16])
{
	auto ce = [](T &a, T &b) { ce_minmax(a, b); };
	batcher_stage<1>(d, ce);
	batcher_stage<2>(d, ce);
	batcher_stage<4>(d, ce);
	batcher_stage<8>(d, ce);
}
template <typename T> __device__ __forceinline__ void merge16_values(T (&d)[16])
{
	auto ce = [](T &a, T &b) { ce_minmax(a, b); };
	batcher_stage<8>(d, ce);
}

// Batcher's odd-even merge sort over `cnt` values held in the LDS, by a whole workgroup (at: index -> LDS word): every comparator
// puts the smaller value at the lower index, so the places behind `cnt` up to the next power of two can be left out (they would
// hold +infinity and never move).  (log2 N)(log2 N + 1) / 2 rounds of at most N / 2 comparators: 91 rounds for 5120 values --
// the bounded way out for a leaf whose bins are too uneven for the placement + register passes (they would need one pass per
// eight values of the fullest bin).
template <int BLOCK, typename T, typename AT> __device__ __forceinline__ void batcher_sort_lds(T *stage, const u32 cnt, AT &&at)
{
	u32 N = 16;
	while (N < cnt)
		N <<= 1;
	for (u32 p = 1; p < N; p <<= 1) {
		for (u32 k = p; k >= 1; k >>= 1) {
			const u32 j0 = k % p;   // (k == p: 0)
			for (u32 idx = threadIdx.x; idx < N / 2; idx += BLOCK) {
				const u32 a = j0 + (idx / k) * 2 * k + idx % k, b = a + k;
				if (b < cnt && a / (2 * p) == b / (2 * p)) {
					const T x = stage[at(a)], y = stage[at(b)];
					if (y < x) {
						stage[at(a)] = y;
						stage[at(b)] = x;
					}
				}
			}
			__syncthreads();
		}
	}
}

template <int BLOCK_, int CAP_, int WPE_, int NBITS_ = 12> struct LeafKCfg {
	static constexpr int BLOCK = BLOCK_, CAP = CAP_, WPE = WPE_, NW = BLOCK_ / 64, NBITS = NBITS_;
	static constexpr int NCH = (CAP / 16 + BLOCK - 1) / BLOCK;   // chunks of sixteen values per lane
	static constexpr int NBIN = 1 << NBITS, NCELLW = NBIN / 2;   // (fewer bins for smaller leaves: rsx_leafp_kernel)
	static constexpr int PLANES = NCELLW / 4 / BLOCK;
	static constexpr u32 MAXBIN = 9, MAXBIN2 = 25;
	// The staged leaf is kept TRANSPOSED: value p lies in row p % 16, column p / 16 of a 16 x S matrix, so that the lanes of a
	// wave -- one chunk of sixteen consecutive values each -- read and write consecutive words (one value per DS instruction,
	// no bank conflicts).  With the values in staging order a lane's chunk is 64 or 128 contiguous bytes, sixteen lanes share
	// four (two) bank groups, and the chunk traffic of 8-byte values cost more than the placement (first version: the u64
	// leaves of cfg 3 1.95 ms against 1.87 with the LDS passes of round 3).
	static constexpr int S = CAP / 16 + 3;   // columns: the chunks + what the shifted pass and the padding reach behind them (odd)
	static_assert(CAP % 16 == 0 && CAP <= 16384, "whole chunks; bin starts fit 16 bits");
	static constexpr u32 PBITS = CAP > 8192 ? 14u : 13u;   // rsx_leafp_kernel: bits of a pair's position in its slot
	static_assert(PLANES == 1 || PLANES == 2, "one or two vectors of cells per thread");
	static_assert(NCELLW == 4 * BLOCK * PLANES, "every cell in some thread's vectors");
	static_assert(S % 2 == 1, "rows that start in different banks");
};

// KT: 8-byte keys; CT: u32 (every column of the leaf in the low word) or u64.  A launch takes the leaves that need its CT.
// redo / SegCtl::nredo / SegCtl::leaf16: as rsx_leaf16_kernel.
// SLOT32: the slots hold the low word of the DERIVED keys (SegCtl::narrow: the level-2 pass wrote four bytes per key); the
// upper word is the same for a whole slot -- key0's, with the slot's two digits where the MSB passes' columns lie above bit 32.
// That form has no list: a leaf with bins too full for the register passes is sorted by batcher_sort_lds.
template <typename KT, typename CT, typename C, bool SLOT32 = false>
__global__ __launch_bounds__(C::BLOCK, C::WPE) void rsx_leafk_kernel(KT *__restrict__ src, KT *__restrict__ aux,
                                                                     const Plan *__restrict__ plan,
                                                                     const LeafSeg *__restrict__ segtab, SegCtl *__restrict__ ctl,
                                                                     KdfArgs<KT> ka, u32 lo, u32 hi, const KT *__restrict__ slots,
                                                                     u32 slack_cap, u32 *__restrict__ redo, u32 maxbin2 = C::MAXBIN2)
{
	static_assert(sizeof(KT) == 8 && (sizeof(CT) == 4 || sizeof(CT) == 8), "8-byte keys carried as 4- or 8-byte values");
	static_assert(C::NBITS >= 9 && C::NBITS <= 12, "bins: a column and the top bits of the next");
	constexpr u32 NB2 = C::NBITS - 8;
	constexpr int BLOCK = C::BLOCK, CAP = C::CAP, NCH = C::NCH, NCELLW = C::NCELLW, NW = C::NW, PLANES = C::PLANES;
	constexpr int NK = (CAP + BLOCK - 1) / BLOCK;   // keys per thread
	const u32 hyb = plan->hyb, ncols = plan->ncols;
	const u32 mode = ctl->mode, maxleaf = ctl->maxleaf, nseg = ctl->nleaf, on = ctl->leaf16;
	if (hyb != HYB_TWO_LEVEL || ncols < 4 || mode != SEG_MODE_LEAVES || maxleaf <= lo || maxleaf > hi || !on)
		return;
	static_assert(!SLOT32 || sizeof(CT) == 4, "four-byte slots, four-byte values");
	if ((ctl->narrow != 0u) != SLOT32)
		return;   // (the other form's sort)
	// the leaf's columns: all kept columns below the two the MSB passes went by; bins from the highest and the one below it
	const u32 c_hi = plan->cols[ncols - 3] & 7u, c_nx = plan->cols[ncols >= 4 ? ncols - 4 : 0] & 7u;
	const bool one_col = ncols - 2 < 2;   // (cannot happen with four kept columns; kept for the shifts below)
	if ((sizeof(CT) == 4) != (c_hi <= 3u))
		return;   // (the other instantiation's leaves)
	const u32 sh_hi = 8 * c_hi, sh_nx = one_col ? 0u : 8 * c_nx + 8 - NB2;
	KT *out = (ncols & 1) ? aux : src;   // radix_sort.hpp:92
	__shared__ __attribute__((aligned(16))) u32 cell[NCELLW + 64];
	constexpr int S = C::S;
	// P6 (values carried as u64): what a leaf sorts by lies below bit 48 -- its columns are kept columns BELOW the two the MSB passes
	// went by, so the highest of them is column 5 at most -- and the staged leaf holds exactly those 48 bits, as a 32-bit plane
	// (bits 16 .. 47) and a 16-bit plane (bits 0 .. 15): 6 bytes per key instead of 8, 40 KB of LDS per 5120-key leaf instead of
	// 50, FOUR workgroups per CU instead of three (tools/ubench/leafk_probe: the leaves of 2^28 u64 keys 1.29 -> 1.13 ms; the
	// phases of a leaf hardly overlap inside one workgroup, so what counts is how many are resident).  In registers a value is
	// the 48 bits in a u64: the networks compare it as before.
	constexpr bool P6 = sizeof(CT) == 8;
	constexpr int S2 = (S + 3) & ~1;   // (the 16-bit plane's row pitch)
	__shared__ __attribute__((aligned(16))) CT stage[P6 ? 1 : 16 * S + 64];   // + a place per lane for values that do not exist
	__shared__ __attribute__((aligned(16))) u32 st_hi[P6 ? 16 * S + 64 : 1];
	__shared__ __attribute__((aligned(16))) unsigned short st_lo[P6 ? 16 * S2 + 64 : 1];
	auto at = [](u32 p) { return (p & 15u) * (u32)S + (p >> 4); };
	// element (row r, column c) of the transposed leaf
	auto put = [&](u32 r, u32 c, CT v) {
		if constexpr (P6) {
			st_hi[r * S + c] = (u32)((u64)v >> 16);
			st_lo[r * S2 + c] = (unsigned short)v;
		} else {
			stage[r * S + c] = v;
		}
	};
	auto get = [&](u32 r, u32 c) -> CT {
		if constexpr (P6)
			return (CT)(((u64)st_hi[r * S + c] << 16) | st_lo[r * S2 + c]);
		else
			return stage[r * S + c];
	};
	auto put_at = [&](u32 p, CT v) { put(p & 15u, p >> 4, v); };
	auto get_at = [&](u32 p) -> CT { return get(p & 15u, p >> 4); };
	constexpr KT LOW48 = (KT)0xFFFFFFFFFFFFull;
	__shared__ u32 ws[NW], wmax[NW];
	const u32 tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
	for (u32 s = blockIdx.x; s < nseg; s += gridDim.x) {
		[&]() {   // (one leaf; see LEAF_ONE_PER_GROUP)
		const LeafSeg ls = segtab[s];
		const u32 cnt = ls.cnt, slot = ls.slot;
		if (cnt == 0)
			return;   // (next leaf)
		const KT *q = slot ? slots + (u64)(slot - 1) * slack_cap : (const KT *)src + ls.beg;
		// the keys, one per lane and round (consecutive lanes read consecutive keys), cut to the carried type
		CT kv[NK];
		KT first;
		if constexpr (SLOT32) {
			const u32 *q32 = (const u32 *)slots + (u64)(slot - 1) * slack_cap;
			// the slot's upper word: the first key's, with the slot's digits at the MSB passes' bit positions
			const u32 sh1 = ctl->shift1, sh2 = ctl->shift2;
			KT up = (KT)(((u64)ctl->key0_hi << 32) | ctl->key0_lo);
			up = (up & ~((KT)0xFFu << sh1)) | ((KT)((slot - 1) >> 8) << sh1);
			up = (up & ~((KT)0xFFu << sh2)) | ((KT)((slot - 1) & 255u) << sh2);
			first = up;
			// (a slot filled by rsx_pass64a_kernel holds its values at both ends: `back` of them in its last LEAF16_BACK places)
			const u32 back = ls.ncols >> 16, front = cnt - back;
#pragma unroll
			for (int j = 0; j < NK; ++j) {
				const u32 e = tid + BLOCK * j;
				const u32 at_e = e < front ? e : slack_cap - LEAF16_BACK + (e - front);   // (one load per value: the place is chosen, not the value)
				kv[j] = e < cnt ? (CT)q32[at_e] : (CT)0;
			}
		} else {
			first = kdf_apply(q[0], ka);
#pragma unroll
			for (int j = 0; j < NK; ++j) {
				const u32 e = tid + BLOCK * j;
				kv[j] = e < cnt ? (CT)(P6 ? (kdf_apply(q[e], ka) & LOW48) : kdf_apply(q[e], ka)) : (CT)0;
			}
		}
		{
			const u32x4 zero = {0, 0, 0, 0};
#pragma unroll
			for (int j = 0; j < PLANES; ++j)
				((u32x4 *)cell)[tid + BLOCK * j] = zero;
		}
		__syncthreads();
		auto cell_of = [&](CT v, bool valid, u32 &sh) -> u32 * {
			const u32 bin = (((u32)(v >> sh_hi) & 0xFFu) << NB2) | ((u32)(v >> sh_nx) & ((1u << NB2) - 1u));
			sh = (bin & 1u) << 4;
			return &cell[valid ? bin >> 1 : NCELLW + lane];
		};
#pragma unroll
		for (int j = 0; j < NK; ++j) {
			if (BLOCK * j < (int)cnt) {
				u32 sh;
				u32 *a = cell_of(kv[j], tid + BLOCK * j < cnt, sh);
				__hip_atomic_fetch_add(a, 1u << sh, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
			}
		}
		__syncthreads();
		u32x4 c[PLANES];
		u32 pk = 0, mxp = 0;
#pragma unroll
		for (int j = 0; j < PLANES; ++j) {
			c[j] = ((const u32x4 *)cell)[tid + BLOCK * j];
			u32 run = 0;
#pragma unroll
			for (int i = 0; i < 4; ++i) {
				const u32 x = c[j][i];
				mxp = pk_max_u16(mxp, x);
				const u32 lo16 = x & 0xFFFFu, hs = run + lo16;
				c[j][i] = run | (hs << 16);
				run = hs + (x >> 16);
			}
			pk |= run << (16 * j);
		}
		u32 mx = (mxp & 0xFFFFu) > (mxp >> 16) ? (mxp & 0xFFFFu) : (mxp >> 16);
		const u32 incl = wave_incl_scan_dpp(pk);
#pragma unroll
		for (int o = 32; o > 0; o >>= 1) {
			const u32 y = (u32)__shfl_xor((int)mx, o);
			mx = mx > y ? mx : y;
		}
		if (lane == 63) {
			ws[wid] = incl;
			wmax[wid] = mx;
		}
		__syncthreads();
		mx = wmax[0];
#pragma unroll
		for (int w = 1; w < NW; ++w)
			mx = mx > wmax[w] ? mx : wmax[w];
		if (!SLOT32 && mx > maxbin2) {
			if (tid == 0)
				redo[atomicAdd(&ctl->nredo, 1u)] = s;
			return;   // (next leaf)
		}
		{
			u32 base = 0, tot = 0;
#pragma unroll
			for (u32 w = 0; w < (u32)NW; ++w) {
				const u32 a = ws[w];
				base += w < wid ? a : 0u;
				tot += a;
			}
			const u32 e = incl - pk + base;
			const u32 o[2] = {e & 0xFFFFu, (tot & 0xFFFFu) + (e >> 16)};
#pragma unroll
			for (int j = 0; j < PLANES; ++j) {
				const u32 bb = o[j] | (o[j] << 16);
				u32x4 x;
#pragma unroll
				for (int i = 0; i < 4; ++i)
					x[i] = c[j][i] + bb;
				((u32x4 *)cell)[tid + BLOCK * j] = x;
			}
		}
		__syncthreads();
#pragma unroll
		for (int j = 0; j < NK; ++j) {
			if (BLOCK * j < (int)cnt) {
				const bool valid = tid + BLOCK * j < cnt;
				u32 sh;
				u32 *a = cell_of(kv[j], valid, sh);
				const u32 old = __hip_atomic_fetch_add(a, 1u << sh, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
				const u32 pos = (old >> sh) & 0xFFFFu;
				if constexpr (P6) {
					st_hi[valid ? at(pos) : 16 * S + lane] = (u32)((u64)kv[j] >> 16);
					st_lo[valid ? (pos & 15u) * (u32)S2 + (pos >> 4) : 16 * S2 + lane] = (unsigned short)kv[j];
				} else {
					stage[valid ? at(pos) : 16 * S + lane] = kv[j];
				}
			}
		}
		if (tid < 32)
			put_at(cnt + tid, P6 ? (CT)LOW48 : (CT)~(CT)0);   // what the last chunks read behind the leaf's end sorts last
		__syncthreads();
		if constexpr (SLOT32) {
			if (mx > maxbin2)
				batcher_sort_lds<BLOCK>(stage, cnt, at);   // (bins too full for the register passes: the network over the whole leaf)
		}
		const u32 npass = (SLOT32 && mx > maxbin2) ? 0u : mx > 17u ? 4u : mx > C::MAXBIN ? 3u : 2u;   // (a bin of up to 9 / 17 / 25 values spans 2 / 3 / 4 blocks of eight)
		for (u32 pass = 0; pass < npass; ++pass) {
			const u32 off = 8 * (pass & 1);
#pragma unroll
			for (int r = 0; r < NCH; ++r) {
				const u32 ch = tid + BLOCK * r;
				if (16 * ch + off < cnt) {
					// value i of the chunk at 16 ch + off: row (i + off) % 16, column ch or ch + 1
					CT d[16];
#pragma unroll
					for (int i = 0; i < 16; ++i)
						d[i] = (pass & 1) ? (i < 8 ? get(i + 8, ch) : get(i - 8, ch + 1)) : get(i, ch);
					if (pass == 0)
						sort16_values(d);
					else
						merge16_values(d);   // (both halves were sorted by the pass before)
#pragma unroll
					for (int i = 0; i < 16; ++i) {
						if (pass & 1) {
							if (i < 8)
								put(i + 8, ch, d[i]);
							else
								put(i - 8, ch + 1, d[i]);
						} else {
							put(i, ch, d[i]);
						}
					}
				}
			}
			__syncthreads();
		}
		{
			constexpr u32 CBITS = 8 * sizeof(CT);
			// what every key of the leaf has above the carried bits (the MSB passes' digits, columns that were skipped)
			const KT upper = P6 ? (KT)(first & ~LOW48) : (KT)(first >> (CBITS & 63) << (CBITS & 63));
			KT *o = out + ls.beg;
			for (u32 i0 = 2 * tid; i0 < cnt; i0 += 2 * BLOCK) {
				KT kk[2];
				kk[0] = kdf_invert((KT)(upper | (KT)get_at(i0)), ka);
				kk[1] = kdf_invert((KT)(upper | (KT)get_at(i0 + 1)), ka);
				if (i0 + 2 <= cnt)
					store_chunk<KT, 2>(o + i0, kk);
				else
					o[i0] = kk[0];
			}
		}
		}();
		if constexpr (LEAF_ONE_PER_GROUP)
			break;
	}
}

// ---- the leaves of 8-byte keys carried as 8-byte values, as a kernel of their own (round 5) -----------------------------------
// rsx_leafk_kernel<KT, u64> above, specialised: the slot's keys with 16-byte loads (two keys per lane: any order will do), the
// staged values in 6 bytes (P6: four workgroups per CU), one leaf per workgroup.  tools/ubench/leafk_probe (2^28 u64 keys,
// 65536 leaves): 1.43 ms as round 4 left it -> 1.29 without the open loop -> 1.13 staged in 6 bytes -> 1.03 with 16-byte
// loads.  (The general kernel keeps the 4-byte-carried and SLOT32 forms.)
struct LeafK8Cfg {
	static constexpr int BLOCK = 512, CAP = 5120, WPE = 8, NW = 8, NBITS = 12;
	static constexpr bool LOOP = false, P6 = true, VLOAD = true;
	static constexpr int SKIP = 0;
	static constexpr int NCH = (CAP / 16 + BLOCK - 1) / BLOCK;
	static constexpr int NBIN = 1 << NBITS, NCELLW = NBIN / 2, NVEC = NCELLW / 4;
	static constexpr int PLANES = (NVEC + BLOCK - 1) / BLOCK;
	static constexpr u32 MAXBIN = 9, MAXBIN2 = 25;
	static constexpr int S = CAP / 16 + 3;
};

template <typename KT, typename CT, typename C>
__global__ __launch_bounds__(C::BLOCK, C::WPE) void rsx_leafk8_kernel(KT *__restrict__ src, KT *__restrict__ aux,
                                                                      const Plan *__restrict__ plan,
                                                                      const LeafSeg *__restrict__ segtab, SegCtl *__restrict__ ctl,
                                                                      KdfArgs<KT> ka, u32 lo, u32 hi, const KT *__restrict__ slots,
                                                                      u32 slack_cap, u32 *__restrict__ redo, u32 maxbin2 = C::MAXBIN2)
{
	static_assert(sizeof(KT) == 8 && (sizeof(CT) == 4 || sizeof(CT) == 8), "8-byte keys carried as 4- or 8-byte values");
	constexpr u32 NB2 = C::NBITS - 8;
	constexpr int BLOCK = C::BLOCK, CAP = C::CAP, NCH = C::NCH, NCELLW = C::NCELLW, NW = C::NW, PLANES = C::PLANES, NVEC = C::NVEC;
	constexpr int NK = (CAP + BLOCK - 1) / BLOCK;
	const u32 hyb = plan->hyb, ncols = plan->ncols;
	const u32 mode = ctl->mode, maxleaf = ctl->maxleaf, nseg = ctl->nleaf, on = ctl->leaf16;
	if (hyb != HYB_TWO_LEVEL || ncols < 4 || mode != SEG_MODE_LEAVES || maxleaf <= lo || maxleaf > hi || !on)
		return;
	if (ctl->narrow != 0u)
		return;
	const u32 c_hi = plan->cols[ncols - 3] & 7u, c_nx = plan->cols[ncols >= 4 ? ncols - 4 : 0] & 7u;
	if ((sizeof(CT) == 4) != (c_hi <= 3u))
		return;
	const u32 sh_hi = 8 * c_hi, sh_nx = 8 * c_nx + 8 - NB2;
	KT *out = (ncols & 1) ? aux : src;
	__shared__ __attribute__((aligned(16))) u32 cell[NCELLW + 64];
	constexpr int S = C::S;
	constexpr bool P6 = C::P6 && sizeof(CT) == 8;
	constexpr int S2 = (S + 3) & ~1;   // (the 16-bit plane's row pitch: an even number of halves, rows start in different banks)
	__shared__ __attribute__((aligned(16))) CT stage[P6 ? 1 : 16 * S + 64];
	__shared__ __attribute__((aligned(16))) u32 st_hi[P6 ? 16 * S + 64 : 1];
	__shared__ __attribute__((aligned(16))) unsigned short st_lo[P6 ? 16 * S2 + 64 : 1];
	auto at = [](u32 p) { return (p & 15u) * (u32)S + (p >> 4); };
	// element (row r, column c) of the staged leaf; the places behind the rows (16 * S + lane) take values that do not exist
	auto put = [&](u32 r, u32 c, CT v) {
		if constexpr (P6) {
			st_hi[r * S + c] = (u32)((u64)v >> 16);
			st_lo[r * S2 + c] = (unsigned short)v;
		} else {
			stage[r * S + c] = v;
		}
	};
	auto get = [&](u32 r, u32 c) -> CT {
		if constexpr (P6)
			return (CT)(((u64)st_hi[r * S + c] << 16) | st_lo[r * S2 + c]);
		else
			return stage[r * S + c];
	};
	auto put_at = [&](u32 p, CT v) { put(p & 15u, p >> 4, v); };
	auto get_at = [&](u32 p) -> CT { return get(p & 15u, p >> 4); };
	__shared__ u32 ws[NW], wmax[NW];
	const u32 tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
	for (u32 s = blockIdx.x; s < nseg; s += gridDim.x) {
		const LeafSeg ls = segtab[s];
		const u32 cnt = ls.cnt, slot = ls.slot;
		if (cnt != 0) {
			const KT *q = slot ? slots + (u64)(slot - 1) * slack_cap : (const KT *)src + ls.beg;
			// the leaf's values in registers.  P6: 48 bits each, kept as the upper 32 (kh) and the lower 16 of two neighbours in one
			// register (kl) -- fifteen registers for ten values instead of twenty: with 64 to a lane (four workgroups per CU) the
			// whole values spilled four of them (20 bytes of scratch per lane, 1.22 x the bytes: profiles/r05/configs, first run)
			CT kv[P6 ? 1 : NK];
			u32 kh[P6 ? NK : 1], kl[P6 ? (NK + 1) / 2 : 1];
			auto set = [&](int j, KT k) {
				if constexpr (P6) {
					kh[j] = (u32)(k >> 16);
					kl[j >> 1] = (j & 1) ? (kl[j >> 1] | ((u32)k << 16)) : ((u32)k & 0xFFFFu);
				} else {
					kv[j] = (CT)k;
				}
			};
			auto val = [&](int j) -> CT {
				if constexpr (P6)
					return (CT)(((u64)kh[j] << 16) | ((kl[j >> 1] >> (16 * (j & 1))) & 0xFFFFu));
				else
					return kv[j];
			};
			const KT first = kdf_apply(q[0], ka);
			// element index of register j (VLOAD: lane t holds elements 2 t, 2 t + 1 of every 2 * BLOCK: a slot starts on a 16-byte
			// boundary and its capacity is even, so the second element of a vector is the slot's own even behind the last key)
			auto elem_of = [&](int j) { return C::VLOAD ? 2u * tid + 2u * BLOCK * (u32)(j >> 1) + (u32)(j & 1) : tid + BLOCK * (u32)j; };
			if constexpr (C::VLOAD) {
				static_assert(NK % 2 == 0, "whole vectors");
				typedef KT kvec_t __attribute__((ext_vector_type(2)));
#pragma unroll
				for (int j = 0; j < NK; j += 2) {
					const u32 e = elem_of(j);
					kvec_t x = {0, 0};
					if (e < cnt)
						x = *(const kvec_t *)(q + e);
					set(j, P6 ? (kdf_apply(x[0], ka) & (KT)0xFFFFFFFFFFFFull) : kdf_apply(x[0], ka));
					set(j + 1, P6 ? (kdf_apply(x[1], ka) & (KT)0xFFFFFFFFFFFFull) : kdf_apply(x[1], ka));
				}
			} else {
#pragma unroll
				for (int j = 0; j < NK; ++j) {
					const u32 e = tid + BLOCK * j;
					set(j, e < cnt ? (P6 ? (kdf_apply(q[e], ka) & (KT)0xFFFFFFFFFFFFull) : kdf_apply(q[e], ka)) : (KT)0);
				}
			}
			{
				const u32x4 zero = {0, 0, 0, 0};
#pragma unroll
				for (int j = 0; j < PLANES; ++j)
					if (tid + BLOCK * j < (u32)NVEC)
						((u32x4 *)cell)[tid + BLOCK * j] = zero;
			}
			__syncthreads();
			auto cell_of = [&](CT v, bool valid, u32 &sh) -> u32 * {
				const u32 bin = (((u32)(v >> sh_hi) & 0xFFu) << NB2) | ((u32)(v >> sh_nx) & ((1u << NB2) - 1u));
				sh = (bin & 1u) << 4;
				return &cell[valid ? bin >> 1 : NCELLW + lane];
			};
			u32 mx = 0;
			bool handed_on = false;
			if constexpr (C::SKIP & 2) {
#pragma unroll
				for (int j = 0; j < NK; ++j)
					if (elem_of(j) < cnt)
						put_at(elem_of(j), val(j));
			} else {
#pragma unroll
				for (int j = 0; j < NK; ++j) {
					if ((C::VLOAD ? 2 * BLOCK * (j >> 1) : BLOCK * j) < (int)cnt) {
						u32 sh;
						u32 *a = cell_of(val(j), elem_of(j) < cnt, sh);
						__hip_atomic_fetch_add(a, 1u << sh, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
					}
				}
				__syncthreads();
				u32x4 c[PLANES];
				u32 pk = 0, mxp = 0;
#pragma unroll
				for (int j = 0; j < PLANES; ++j) {
					c[j] = u32x4{0, 0, 0, 0};
					if (tid + BLOCK * j < (u32)NVEC)
						c[j] = ((const u32x4 *)cell)[tid + BLOCK * j];
					u32 run = 0;
#pragma unroll
					for (int i = 0; i < 4; ++i) {
						const u32 x = c[j][i];
						mxp = pk_max_u16(mxp, x);
						const u32 lo16 = x & 0xFFFFu, hs = run + lo16;
						c[j][i] = run | (hs << 16);
						run = hs + (x >> 16);
					}
					pk |= run << (16 * j);
				}
				mx = (mxp & 0xFFFFu) > (mxp >> 16) ? (mxp & 0xFFFFu) : (mxp >> 16);
				const u32 incl = wave_incl_scan_dpp(pk);
#pragma unroll
				for (int o = 32; o > 0; o >>= 1) {
					const u32 y = (u32)__shfl_xor((int)mx, o);
					mx = mx > y ? mx : y;
				}
				if (lane == 63) {
					ws[wid] = incl;
					wmax[wid] = mx;
				}
				__syncthreads();
				mx = wmax[0];
#pragma unroll
				for (int w = 1; w < NW; ++w)
					mx = mx > wmax[w] ? mx : wmax[w];
				if (mx > maxbin2) {
					if (tid == 0)
						redo[atomicAdd(&ctl->nredo, 1u)] = s;
					handed_on = true;
				}
				if (!handed_on) {
					u32 base = 0, tot = 0;
#pragma unroll
					for (u32 w = 0; w < (u32)NW; ++w) {
						const u32 a = ws[w];
						base += w < wid ? a : 0u;
						tot += a;
					}
					const u32 e = incl - pk + base;
					const u32 o[2] = {e & 0xFFFFu, (tot & 0xFFFFu) + (e >> 16)};
#pragma unroll
					for (int j = 0; j < PLANES; ++j) {
						const u32 bb = o[j] | (o[j] << 16);
						u32x4 x;
#pragma unroll
						for (int i = 0; i < 4; ++i)
							x[i] = c[j][i] + bb;
						if (tid + BLOCK * j < (u32)NVEC)
							((u32x4 *)cell)[tid + BLOCK * j] = x;
					}
					__syncthreads();
#pragma unroll
					for (int j = 0; j < NK; ++j) {
						if ((C::VLOAD ? 2 * BLOCK * (j >> 1) : BLOCK * j) < (int)cnt) {
							const bool valid = elem_of(j) < cnt;
							u32 sh;
							u32 *a = cell_of(val(j), valid, sh);
							const u32 old = __hip_atomic_fetch_add(a, 1u << sh, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
							const u32 pos = (old >> sh) & 0xFFFFu;
							if constexpr (P6) {
								st_hi[valid ? at(pos) : 16 * S + lane] = kh[j];
								st_lo[valid ? (pos & 15u) * (u32)S2 + (pos >> 4) : 16 * S2 + lane] = (unsigned short)(kl[j >> 1] >> (16 * (j & 1)));
							} else {
								stage[valid ? at(pos) : 16 * S + lane] = kv[j];
							}
						}
					}
				}
			}
			if (!handed_on) {
				if (tid < 32)
					put_at(cnt + tid, P6 ? (CT)0xFFFFFFFFFFFFull : (CT)~(CT)0);
				__syncthreads();
				if constexpr (!(C::SKIP & 1)) {
					const u32 npass = mx > 17u ? 4u : mx > C::MAXBIN ? 3u : 2u;
					for (u32 pass = 0; pass < npass; ++pass) {
#pragma unroll
						for (int r = 0; r < NCH; ++r) {
							const u32 ch = tid + BLOCK * r;
							const u32 off = 8 * (pass & 1);
							if (16 * ch + off < cnt) {
								CT d[16];
#pragma unroll
								for (int i = 0; i < 16; ++i)
									d[i] = (pass & 1) ? (i < 8 ? get(i + 8, ch) : get(i - 8, ch + 1)) : get(i, ch);
								if (pass == 0)
									sort16_values(d);
								else
									merge16_values(d);
#pragma unroll
								for (int i = 0; i < 16; ++i)
									if (pass & 1) {
										if (i < 8)
											put(i + 8, ch, d[i]);
										else
											put(i - 8, ch + 1, d[i]);
									} else {
										put(i, ch, d[i]);
									}
							}
						}
						__syncthreads();
					}
				}
				if constexpr (!(C::SKIP & 4)) {
					constexpr u32 CBITS = 8 * sizeof(CT);
					const KT upper = P6 ? (KT)(first >> 48 << 48) : sizeof(CT) == 8 ? (KT)0 : (KT)(first >> (CBITS & 63) << (CBITS & 63));
					KT *o = out + ls.beg;
					for (u32 i0 = 2 * tid; i0 < cnt; i0 += 2 * BLOCK) {
						KT kk[2];
						kk[0] = kdf_invert((KT)(upper | (KT)get_at(i0)), ka);
						kk[1] = kdf_invert((KT)(upper | (KT)get_at(i0 + 1)), ka);
						if (i0 + 2 <= cnt)
							store_chunk<KT, 2>(o + i0, kk);
						else
							o[i0] = kk[0];
					}
				}
			}
		}
		if constexpr (!C::LOOP)
			break;
		__syncthreads();
	}
}


// ---- leaves of key + payload and rank sorts (4-byte keys, 4-byte payloads: BASELINE.json's cfg 4) --------------------------
// These leaves must be STABLE (equal keys keep their payloads' order, radix_sort_rank.hpp:82-90).  A slot's pairs lie in the
// order the two (stable) MSB passes brought them, so position i in the slot is the tie-break: the leaf sorts the 32-bit
// compounds (low sixteen bits of the derived key << 16 | i), which are all different -- any sorting method sorts them the
// stable way.  The payloads are staged where they lie (linear) and gathered through i at the end; the compounds go through
// the placement by the key's top twelve bits and the register passes of rsx_leafk_kernel.  Four data-dependent LDS
// operations per pair (count, cursor, staging store, payload gather) on 4-byte words instead of six on 8-byte ones
// (rsx_leaf_pairs_kernel carries the pair as one 8-byte value through two LDS passes).
// K16 (sorts without a histogram, round 6): the key slots hold the low two bytes of what the level-2 pass read -- the DERIVED
// key's, or the packed key's -- which is all a leaf ever took from a key: `kslots` points to 2-byte values, a slot's keys are
// 2 x slack_cap bytes apart (rsx_scatter2_kernel with KTO = u16; 16 -> 14 bytes per pair through the level-2 pass, 8 -> 6 into
// the leaves, and the key slots of 2^28 pairs 0.63 instead of 1.25 GiB).
template <typename KT, typename VT, typename C, bool K16 = false>
__global__ __launch_bounds__(C::BLOCK, C::WPE) void rsx_leafp_kernel(const KT *__restrict__ kslots, const VT *__restrict__ vslots,
                                                                     u32 slack_cap, KT *__restrict__ kout, VT *__restrict__ vout,
                                                                     const Plan *__restrict__ plan,
                                                                     const LeafSeg *__restrict__ segtab, SegCtl *__restrict__ ctl,
                                                                     KdfArgs<KT> ka, u32 *__restrict__ redo, u32 maxbin2 = C::MAXBIN2)
{
	static_assert(sizeof(KT) == 4 && sizeof(VT) == 4, "pairs of 4-byte keys and 4-byte payloads");
	constexpr int BLOCK = C::BLOCK, CAP = C::CAP, NCH = C::NCH, NCELLW = C::NCELLW, NW = C::NW, PLANES = C::PLANES, S = C::S;
	constexpr int NV = (CAP / 4 + BLOCK - 1) / BLOCK;   // 16-byte vectors of four keys (and of four payloads) per thread
	const u32 hyb = plan->hyb, ncols = plan->ncols;
	const u32 mode = ctl->mode, nseg = ctl->nleaf, on = ctl->leaf16, maxleaf = ctl->maxleaf;
	if (hyb != HYB_TWO_LEVEL || ncols != 4 || mode != SEG_MODE_LEAVES || !on || maxleaf > (u32)CAP)
		return;
	// SegCtl::compact (rsx_hybrid.hpp): the slots hold PACKED keys -- plain unsigned, their own derived keys -- of which the low
	// LB = SegCtl::shift2 bits are the leaf's to sort by (16 without compaction: the two low byte columns).  The compound is the
	// leaf's bits above the pair's position in the slot, left-aligned: its top twelve bits -- the bins -- then spread over the
	// leaf's bits and, where those are fewer than twelve, the top bits of the position.
	const u32 LB = ctl->compact ? ctl->shift2 : 16u;
	if (ctl->compact)
		ka.fmask = ka.sflip = ka.desc = 0;
	const u32 shk = 32u - LB, shi = (32u - C::PBITS) - LB;   // (the position's thirteen -- slots beyond 8192 pairs: fourteen -- bits right below the leaf's: no unused bit between them dilutes the bins)
	__shared__ __attribute__((aligned(16))) u32 cell[NCELLW + 64];
	__shared__ __attribute__((aligned(16))) u32 stage[16 * S + 64];   // the compounds, transposed (LeafKCfg::S)
	__shared__ __attribute__((aligned(16))) VT pay[CAP];
	__shared__ u32 ws[NW], wmax[NW];
	auto at = [](u32 p) { return (p & 15u) * (u32)S + (p >> 4); };
	const u32 tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
	for (u32 s = blockIdx.x; s < nseg; s += gridDim.x) {
		[&]() {   // (one leaf; see LEAF_ONE_PER_GROUP)
		const LeafSeg ls = segtab[s];
		const u32 cnt = ls.cnt, slot = ls.slot;
		if (cnt == 0)
			return;   // (next leaf)
		const KT *kp = kslots + (u64)(slot - 1) * slack_cap;
		const unsigned short *kp16 = (const unsigned short *)kslots + (u64)(slot - 1) * slack_cap;
		const VT *vp = vslots + (u64)(slot - 1) * slack_cap;
		u32x4 kv[NV], vv[NV];
#pragma unroll
		for (int j = 0; j < NV; ++j) {
			const u32 e0 = 4 * (tid + BLOCK * j);
			kv[j] = u32x4{0, 0, 0, 0};
			vv[j] = u32x4{0, 0, 0, 0};
			if (e0 < cnt) {   // (a slot's capacity is a multiple of 256 pairs: the vector behind the last pair is the slot's own)
				if constexpr (K16) {
					const u32x2 h = *(const u32x2 *)(kp16 + e0);
					kv[j] = u32x4{h[0] & 0xFFFFu, h[0] >> 16, h[1] & 0xFFFFu, h[1] >> 16};
				} else {
					kv[j] = *(const u32x4 *)(kp + e0);
				}
				vv[j] = *(const u32x4 *)(vp + e0);
			}
		}
		{
			const u32x4 zero = {0, 0, 0, 0};
#pragma unroll
			for (int j = 0; j < PLANES; ++j)
				((u32x4 *)cell)[tid + BLOCK * j] = zero;
		}
		// the compounds: low half of the derived key above, position in the slot below; the payloads staged where they lie
#pragma unroll
		for (int j = 0; j < NV; ++j) {
			const u32 e0 = 4 * (tid + BLOCK * j);
			if (e0 < cnt)
				*(u32x4 *)&pay[e0] = vv[j];
#pragma unroll
			for (int e = 0; e < 4; ++e)
				kv[j][e] = ((K16 ? kv[j][e] : (u32)kdf_apply((KT)kv[j][e], ka)) << shk) | ((e0 + e) << shi);
		}
		__syncthreads();
		auto cell_of = [&](u32 c, bool valid, u32 &sh) -> u32 * {
			constexpr int B = 32 - C::NBITS;   // bin = c >> B: word c >> (B + 1), half bit B
			sh = (c >> (B - 4)) & 16u;
			return &cell[valid ? c >> (B + 1) : NCELLW + lane];
		};
#pragma unroll
		for (int j = 0; j < NV; ++j) {
			if (4 * BLOCK * j < (int)cnt) {
#pragma unroll
				for (int e = 0; e < 4; ++e) {
					u32 sh;
					u32 *a = cell_of(kv[j][e], 4 * (tid + BLOCK * j) + e < cnt, sh);
					__hip_atomic_fetch_add(a, 1u << sh, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
				}
			}
		}
		__syncthreads();
		u32x4 c[PLANES];
		u32 pk = 0, mxp = 0;
#pragma unroll
		for (int j = 0; j < PLANES; ++j) {
			c[j] = ((const u32x4 *)cell)[tid + BLOCK * j];
			u32 run = 0;
#pragma unroll
			for (int i = 0; i < 4; ++i) {
				const u32 x = c[j][i];
				mxp = pk_max_u16(mxp, x);
				const u32 lo16 = x & 0xFFFFu, hs = run + lo16;
				c[j][i] = run | (hs << 16);
				run = hs + (x >> 16);
			}
			pk |= run << (16 * j);
		}
		u32 mx = (mxp & 0xFFFFu) > (mxp >> 16) ? (mxp & 0xFFFFu) : (mxp >> 16);
		const u32 incl = wave_incl_scan_dpp(pk);
#pragma unroll
		for (int o = 32; o > 0; o >>= 1) {
			const u32 y = (u32)__shfl_xor((int)mx, o);
			mx = mx > y ? mx : y;
		}
		if (lane == 63) {
			ws[wid] = incl;
			wmax[wid] = mx;
		}
		__syncthreads();
		mx = wmax[0];
#pragma unroll
		for (int w = 1; w < NW; ++w)
			mx = mx > wmax[w] ? mx : wmax[w];
		if (mx > maxbin2) {   // (keys with many duplicates: the LDS passes of rsx_leaf_pairs_kernel)
			if (tid == 0)
				redo[atomicAdd(&ctl->nredo, 1u)] = s;
			return;   // (next leaf)
		}
		{
			u32 base = 0, tot = 0;
#pragma unroll
			for (u32 w = 0; w < (u32)NW; ++w) {
				const u32 a = ws[w];
				base += w < wid ? a : 0u;
				tot += a;
			}
			const u32 e = incl - pk + base;
			const u32 o[2] = {e & 0xFFFFu, (tot & 0xFFFFu) + (e >> 16)};
#pragma unroll
			for (int j = 0; j < PLANES; ++j) {
				const u32 bb = o[j] | (o[j] << 16);
				u32x4 x;
#pragma unroll
				for (int i = 0; i < 4; ++i)
					x[i] = c[j][i] + bb;
				((u32x4 *)cell)[tid + BLOCK * j] = x;
			}
		}
		__syncthreads();
#pragma unroll
		for (int j = 0; j < NV; ++j) {
			if (4 * BLOCK * j < (int)cnt) {
#pragma unroll
				for (int e = 0; e < 4; ++e) {
					const bool valid = 4 * (tid + BLOCK * j) + e < cnt;
					u32 sh;
					u32 *a = cell_of(kv[j][e], valid, sh);
					const u32 old = __hip_atomic_fetch_add(a, 1u << sh, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
					stage[valid ? at((old >> sh) & 0xFFFFu) : 16 * S + lane] = kv[j][e];
				}
			}
		}
		if (tid < 32)
			stage[at(cnt + tid)] = ~0u;
		__syncthreads();
		// (a bin of up to 9 / 17 / 25 compounds spans 2 / 3 / 4 blocks of eight: as many passes.  Packed keys with few bits left for
		// the leaves -- cfg 4 (iii): four -- fill their bins from 32 neighbouring positions each, 10-11 in the fullest: three passes)
		const u32 npass = mx > 17u ? 4u : mx > C::MAXBIN ? 3u : 2u;
		for (u32 pass = 0; pass < npass; ++pass) {
			const u32 off = 8 * (pass & 1);
#pragma unroll
			for (int r = 0; r < NCH; ++r) {
				const u32 ch = tid + BLOCK * r;
				if (16 * ch + off < cnt) {
					u32 d[16];
#pragma unroll
					for (int i = 0; i < 16; ++i)
						d[i] = stage[(pass & 1) ? (i < 8 ? (i + 8) * S + ch : (i - 8) * S + ch + 1) : i * S + ch];
					if (pass == 0)
						sort16_values(d);
					else
						merge16_values(d);
#pragma unroll
					for (int i = 0; i < 16; ++i)
						stage[(pass & 1) ? (i < 8 ? (i + 8) * S + ch : (i - 8) * S + ch + 1) : i * S + ch] = d[i];
				}
			}
			__syncthreads();
		}
		{
			const KT upper = (KT)(((KT)((slot - 1) >> 8) << 24) | ((KT)((slot - 1) & 255u) << 16));
			VT *vo = vout + ls.beg;
			KT *ko = kout ? kout + ls.beg : nullptr;
			for (u32 i0 = 4 * tid; i0 < cnt; i0 += 4 * BLOCK) {
				VT pv[4];
				KT kk[4];
#pragma unroll
				for (int e = 0; e < 4; ++e) {
					const u32 x = stage[at(i0 + e)];
					const u32 pi = (x >> shi) & ((1u << C::PBITS) - 1u);   // (the pair's position in the slot)
					pv[e] = pay[pi < (u32)CAP ? pi : 0u];                  // (behind the leaf's end: padding)
					kk[e] = kdf_invert((KT)(upper | (x >> 16)), ka);       // (keys out: key + payload sorts, never packed)
				}
				if (i0 + 4 <= cnt) {
					store_chunk<VT, 4>(vo + i0, pv);
					if (ko)
						store_chunk<KT, 4>(ko + i0, kk);
				} else {
#pragma unroll
					for (int e = 0; e < 4; ++e) {
						if (i0 + e < cnt) {
							vo[i0 + e] = pv[e];
							if (ko)
								ko[i0 + e] = kk[e];
						}
					}
				}
			}
		}
		__syncthreads();   // (the payloads are staged again before the next leaf's first barrier)
		}();
		if constexpr (LEAF_ONE_PER_GROUP)
			break;
	}
}

}  // namespace rsx
