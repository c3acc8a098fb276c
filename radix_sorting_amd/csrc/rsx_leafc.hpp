// rsx_leafc.hpp -- the leaves of a keys-only two-level sort of 4-byte keys whose two-byte slots hold MORE values than
// rsx_leaf16_kernel's shapes take (rsx_leaf16.hpp: 5120): arrays from about 2^28 keys up to 2^31, gfx950.
//
// What such a leaf has to do is what rsx_leaf16.hpp says: a slot holds up to `slack_cap` 16-bit values, the low half of the derived
// keys of one (digit, digit) bucket in any order; the result wants them ascending, widened to the caller's elements -- the
// reference's last two passes (radix_sort.hpp:82-90, columns 0 and 1).  Keys that compare equal are the same bits, so nothing
// has to be stable and nothing has to MOVE: with 8 Ki .. 40 Ki values of sixteen bits a leaf is a COUNTING sort --
//
//   1. count: one LDS atomic per value on a 16-bit cell per VALUE (65536 cells, 128 KiB: the leaf's whole histogram);
//   2. scan: thread t owns the cells of the values 64 t .. 64 t + 63 (read without bank conflicts: the words are stored with
//      three address bits XORed, see `cell_word`), sums them, one scan over the workgroup gives every thread the place of its
//      first value in the leaf;
//   3. mark: the cells are in registers now and their LDS becomes the leaf's staging area, one 16-bit place per output position,
//      zeroed; every thread puts each value it owns that occurs at the place of its FIRST occurrence;
//   4. write out: a place holds the last mark at or before it -- marks ascend with the places, so that is a running MAXIMUM: four
//      places per lane, a DPP max-scan over the wave's 256 places, the largest mark of every block of 256 through the LDS for
//      the blocks behind it, the upper half from the slot's digits, kdf_invert, 16-byte stores.
//
// Per value: one LDS atomic, about a dozen VALU instructions in steps 2-3 and as many in step 4 -- against six data-dependent
// LDS operations per value and column in rsx_leaf_sort_kernel's two stable passes.  Nothing depends on how the values are spread:
// apart from the atomics' conflicts (a leaf of 40000 equal values: 64 lanes on one word) a leaf costs what any other costs,
// and there is no list of leaves left over.
// One workgroup per CU (the histogram fills the LDS), persistent: a workgroup takes every gridDim.x-th leaf and requests the next
// leaf's values as soon as the current ones are counted.
#pragma once

#include "rsx_leaf16.hpp"

namespace rsx {

template <int NV_> struct LeafCCfg {
	static constexpr int BLOCK = 1024, NW = BLOCK / 64, WPE = 4;
	static constexpr int NV = NV_;                   // 16-byte vectors of eight values per thread
	static constexpr int CAP = NV * 8 * BLOCK;       // values per leaf (5: 40960 -- the slots of 2^31 keys)
	static constexpr int NCELLW = 32768;             // two 16-bit cells to a word
	static constexpr int NBLK = CAP / 256;           // blocks of 256 places
	static constexpr int NBI = (NBLK + NW - 1) / NW; // ... per wave
	static constexpr int DUMMY = 53248;              // places (16-bit) nobody reads: BLOCK of them, behind the leaf's
	static constexpr int BMAX = 30720;               // word: the blocks' largest marks
	static_assert(CAP + 256 <= DUMMY && (DUMMY + BLOCK) * 2 <= BMAX * 4 && BMAX + NBLK <= NCELLW && NBLK <= 192,
	              "a place per value in what the cells occupied, then the dummies, then the blocks' maxima; 16-bit counts");
};

// inclusive running maximum over the 64 lanes of a wave (as wave_incl_scan_dpp; 0 is what a lane without a neighbour sees)
__device__ __forceinline__ u32 wave_incl_max_dpp(u32 x)
{
	auto mx = [](u32 a, u32 b) { return a > b ? a : b; };
	x = mx(x, (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x111 /* row_shr:1 */, 0xF, 0xF, true));
	x = mx(x, (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x112 /* row_shr:2 */, 0xF, 0xF, true));
	x = mx(x, (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x114 /* row_shr:4 */, 0xF, 0xF, true));
	x = mx(x, (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x118 /* row_shr:8 */, 0xF, 0xF, true));
	x = mx(x, (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x142 /* row_bcast:15 */, 0xA, 0xF, false));
	x = mx(x, (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x143 /* row_bcast:31 */, 0xC, 0xF, false));
	return x;
}
// lane l gets lane l - 1's value, lane 0 gets 0
__device__ __forceinline__ u32 from_prev_lane_or_zero(u32 x)
{
	return (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x138 /* wave_shr:1 */, 0xF, 0xF, true);
}

// segtab / slots / slack_cap / lo / hi: as rsx_leaf16_kernel (the leaves of a two-level plan whose largest lies in (lo, hi]).
// redo != nullptr: the launch BEHIND rsx_leaf16_kernel in one of its larger shapes (slots of 5121 .. 20480 values, where its
// register passes are faster than counting while the values spread: tools/ubench/leafc_probe) -- only the leaves that kernel's
// list names (SegCtl::nredo of them; it has moved their slots' last values up behind the others), or every leaf if it was told
// to stay away (SegCtl::leaf16 == 0: the sample saw the low sixteen bits cluster).
template <typename KT, typename C>
__global__ __launch_bounds__(C::BLOCK, C::WPE) void rsx_leafc_kernel(KT *__restrict__ src, KT *__restrict__ aux,
                                                                     const Plan *__restrict__ plan,
                                                                     const LeafSeg *__restrict__ segtab, const SegCtl *__restrict__ ctl,
                                                                     KdfArgs<KT> ka, u32 lo, u32 hi,
                                                                     const uint16_t *__restrict__ slots, u32 slack_cap,
                                                                     const u32 *__restrict__ redo = nullptr)
{
	static_assert(sizeof(KT) == 4, "4-byte keys: two MSB digits in the slot, two bytes in the leaf");
	constexpr int BLOCK = C::BLOCK, NV = C::NV, NCELLW = C::NCELLW, NW = C::NW;
	const u32 hyb = plan->hyb, ncols = plan->ncols;
	const u32 sh1 = ctl->shift1, sh2 = ctl->shift2;   // the MSB digits' bit positions (24 and 16 unless the keys' top bits are constant)
	const u32 mode = ctl->mode, maxleaf = ctl->maxleaf;
	if (hyb != HYB_TWO_LEVEL || ncols != 4 || mode != SEG_MODE_LEAVES || maxleaf <= lo || maxleaf > hi)
		return;
	const bool listed = redo != nullptr && ctl->leaf16 != 0;
	const u32 nseg = listed ? ctl->nredo : ctl->nleaf;
	auto leaf = [&](u32 i) {
		LeafSeg l = segtab[listed ? redo[i] : i];
		if (listed)
			l.ncols &= 0xFFFFu;   // (a listed leaf's values are all in front)
		return l;
	};
	KT *out = src;   // (four kept columns: the reference's passes end in src, radix_sort.hpp:92)
	(void)aux;
	// cell of value v: half (v & 1) of word v >> 1 -- stored at cell_word(v >> 1); later the staging area (a 16-bit place per position)
	__shared__ __attribute__((aligned(16))) u32 cell[NCELLW + 64];   // + a word per lane for values that do not exist
	__shared__ u32 ws[NW];
	uint16_t *const stage = (uint16_t *)cell;
	u32 *const bmax = cell + C::BMAX;
	// thread t owns words 32 t .. 32 t + 31 and reads them as eight 16-byte vectors: vector j of thread t is stored as vector
	// j ^ (t & 7) of its eight, so that eight neighbouring lanes touch eight different vectors of a 128-byte row of the banks
	auto cell_word = [](u32 w) { return w ^ (((w >> 5) & 7u) << 2); };
	const u32 tid0 = threadIdx.x;
	const KT key0 = (KT)ctl->key0_lo;
	const KT above = sh1 + 8 >= 32u ? (KT)0 : (KT)(key0 >> (sh1 + 8) << (sh1 + 8));   // what every key has above the level-1 digit

	u32x4 kv[NV];
	// the values of leaf `ls`, 16 bytes per lane and step: the front's vectors, then the back's (a slot filled by
	// rsx_pass16a_kernel holds its values at both ends, rsx_leaf16_kernel)
	auto request = [&](const LeafSeg ls, const u32 tid) {
		const u32 back = ls.ncols >> 16, front = ls.cnt - back;
		const uint16_t *q = slots + (u64)(ls.slot - 1) * slack_cap;
		const u32 VF = (front + 7u) >> 3;
#pragma unroll
		for (int j = 0; j < NV; ++j) {
			const u32 v = tid + BLOCK * j;
			const bool isback = v >= VF;
			const u32 e0 = 8 * (isback ? v - VF : v);
			const int left = (int)(isback ? back : front) - (int)e0;
			kv[j] = u32x4{0, 0, 0, 0};
			if (left > 0)
				kv[j] = *(const u32x4 *)(q + (isback ? slack_cap - LEAF16_BACK : 0u) + e0);
		}
	};
	u32 s = blockIdx.x;
	while (s < nseg && leaf(s).cnt == 0)
		s += gridDim.x;
	if (s >= nseg)
		return;
	LeafSeg ls = leaf(s);
	request(ls, tid0);
	for (;;) {
		// (everything a leaf derives from the thread index is derived from an opaque copy of it, made per leaf: as loop invariants
		// the addresses would be hoisted in front of the loop and spilled there -- rsx_pass32.hpp)
		u32 tid = tid0;
		asm volatile("" : "+v"(tid));
		const u32 lane = tid & 63, wid = tid >> 6;
		const u32 cnt = ls.cnt, slot = ls.slot;
		const u32 back = ls.ncols >> 16, front = cnt - back;
		const u32 VF = (front + 7u) >> 3;
		// ---- all cells zero (the leaf before has been written out: the barrier at the loop's end)
		{
			const u32x4 zero = {0, 0, 0, 0};
#pragma unroll
			for (int j = 0; j < NCELLW / 4 / BLOCK; ++j)
				((u32x4 *)cell)[tid + BLOCK * j] = zero;
		}
		__syncthreads();
		// ---- count (a value that does not exist counts in the lane's own word behind the cells)
#pragma unroll
		for (int j = 0; j < NV; ++j) {
			const u32 v = tid + BLOCK * j;
			const bool isback = v >= VF;
			const u32 e0 = 8 * (isback ? v - VF : v);
			const int left = (int)(isback ? back : front) - (int)e0;
			if (__all(left >= 8)) {
#pragma unroll
				for (int k = 0; k < 8; ++k) {
					const u32 val = __builtin_amdgcn_ubfe(kv[j][k >> 1], 16u * (u32)(k & 1), sh2);
					__hip_atomic_fetch_add(&cell[cell_word(val >> 1)], 1u + (val & 1u) * 0xFFFFu, __ATOMIC_RELAXED,
					                       __HIP_MEMORY_SCOPE_WORKGROUP);
				}
			} else if (__any(left > 0)) {
#pragma unroll
				for (int k = 0; k < 8; ++k) {
					const u32 val = __builtin_amdgcn_ubfe(kv[j][k >> 1], 16u * (u32)(k & 1), sh2);
					const u32 w = k < left ? cell_word(val >> 1) : (u32)NCELLW + lane;
					__hip_atomic_fetch_add(&cell[w], 1u + (val & 1u) * 0xFFFFu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
				}
			}
		}
		// the next leaf's values are requested now: the registers are free, and the values cross the memory system while this leaf
		// is scanned, marked and written out
		u32 snext = s + gridDim.x;
		while (snext < nseg && leaf(snext).cnt == 0)
			snext += gridDim.x;
		LeafSeg lnext = ls;
		if (snext < nseg) {
			lnext = leaf(snext);
			request(lnext, tid);
		}
		__syncthreads();
		// ---- scan: the thread's 64 cells into registers, their sum, the place of its first value
		u32 c[32];
		u32 sum = 0;
#pragma unroll
		for (int j = 0; j < 8; ++j) {
			const u32x4 x = ((const u32x4 *)cell)[8 * tid + ((u32)j ^ (tid & 7u))];
#pragma unroll
			for (int i = 0; i < 4; ++i) {
				c[4 * j + i] = x[i];
				sum += (x[i] & 0xFFFFu) + (x[i] >> 16);
			}
		}
		// (the marks take the words apart again: sixty-four counts kept as such would be thirty-two registers more)
#pragma unroll
		for (int w = 0; w < 32; ++w)
			asm volatile("" : "+v"(c[w]));
		const u32 incl = wave_incl_scan_dpp(sum);
		if (lane == 63)
			ws[wid] = incl;
		__syncthreads();   // (and every cell has been read: their LDS is the staging area from here on)
		u32 pos = incl - sum;
#pragma unroll
		for (u32 w = 0; w < (u32)NW; ++w)
			pos += w < wid ? ws[w] : 0u;
		{
			const u32x4 zero = {0, 0, 0, 0};
			const u32 nvec = (cnt + 256u + 7u) >> 3;   // (a wave's last 256 places may lie behind the leaf's end)
			for (u32 v = tid; v < nvec; v += BLOCK)
				((u32x4 *)cell)[v] = zero;
		}
		__syncthreads();
		// ---- mark (no branches: a value that does not occur puts its mark on a place of the thread's own behind everything)
		{
			const u32 dummy = (u32)C::DUMMY + tid;
#pragma unroll
			for (int w = 0; w < 32; ++w) {
				const u32 n0 = c[w] & 0xFFFFu, n1 = c[w] >> 16;
				const u32 v0 = 64u * tid + 2u * (u32)w;
				stage[n0 ? pos : dummy] = (uint16_t)v0;
				pos += n0;
				stage[n1 ? pos : dummy] = (uint16_t)(v0 + 1u);
				pos += n1;
				asm volatile("" : "+v"(pos));   // (one word's marks after the other: nothing of the later words is computed early)
			}
		}
		__syncthreads();
		// ---- write out.  A place's value is the largest mark at or before it.  Wave `wid` takes the blocks of 256 places
		// wid + NW i, four places per lane: first the running maximum inside every block (kept in registers) and the block's
		// largest mark, then -- all blocks' maxima are known -- the keys: the upper half from the slot's digits, kdf_invert,
		// 16-byte stores
		{
			constexpr int NBI = C::NBI;
			uint2 xs[NBI];
			u32 inc[NBI];
			const u32 swid = (u32)__builtin_amdgcn_readfirstlane((int)wid);
			const u32 nblk = (cnt + 255u) >> 8;
#pragma unroll
			for (int i = 0; i < NBI; ++i) {
				const u32 b = swid + (u32)NW * (u32)i;
				if (b < nblk) {
					xs[i] = *(const uint2 *)&stage[256u * b + 4u * lane];
					u32 m = xs[i].x & 0xFFFFu;
					m = m > (xs[i].x >> 16) ? m : xs[i].x >> 16;
					m = m > (xs[i].y & 0xFFFFu) ? m : xs[i].y & 0xFFFFu;
					m = m > (xs[i].y >> 16) ? m : xs[i].y >> 16;
					inc[i] = wave_incl_max_dpp(m);
					if (lane == 63)
						bmax[b] = inc[i];
				}
			}
			__syncthreads();
			// the largest mark in the blocks 0 .. b, for every block b: lane l of every wave holds blocks l, 64 + l, 128 + l
			u32 g[3];
#pragma unroll
			for (int k = 0; k < 3; ++k) {
				const u32 b = 64u * (u32)k + lane;
				g[k] = wave_incl_max_dpp(b < nblk ? bmax[b] : 0u);
				if (k) {
					const u32 before = (u32)__builtin_amdgcn_readlane((int)g[k - 1], 63);
					g[k] = g[k] > before ? g[k] : before;
				}
			}
			const KT upper = (KT)(above | ((KT)((slot - 1) >> 8) << sh1) | ((KT)((slot - 1) & 255u) << sh2));
			KT *o = out + ls.beg;
#pragma unroll
			for (int i = 0; i < NBI; ++i) {
				const u32 b = swid + (u32)NW * (u32)i;
				if (b < nblk) {
					u32 carry = from_prev_lane_or_zero(inc[i]);
					if (b) {
						const u32 bb = b - 1u;
						const u32 gsel = bb < 64u ? g[0] : bb < 128u ? g[1] : g[2];
						const u32 cb = (u32)__builtin_amdgcn_readlane((int)gsel, (int)(bb & 63u));
						carry = carry > cb ? carry : cb;
					}
					u32 m[4];
					m[0] = xs[i].x & 0xFFFFu;
					m[1] = xs[i].x >> 16;
					m[2] = xs[i].y & 0xFFFFu;
					m[3] = xs[i].y >> 16;
					m[0] = m[0] > carry ? m[0] : carry;
					m[1] = m[1] > m[0] ? m[1] : m[0];
					m[2] = m[2] > m[1] ? m[2] : m[1];
					m[3] = m[3] > m[2] ? m[3] : m[2];
					const u32 i0 = 256u * b + 4u * lane;
					KT kk[4];
#pragma unroll
					for (int e = 0; e < 4; ++e)
						kk[e] = kdf_invert((KT)(upper | (KT)m[e]), ka);
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
		}
		if (snext >= nseg)
			break;
		s = snext;
		ls = lnext;
		__syncthreads();   // the staging area has been read before it is zeroed
	}
}

}   // namespace rsx
