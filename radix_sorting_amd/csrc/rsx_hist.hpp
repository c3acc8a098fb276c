// rsx_hist.hpp -- loop 1 of rs_sort_main (radix_sort.hpp:47-58) on gfx950: every 8-bit column's histogram in ONE read
// of the keys + the pre-sorted test.  (Included by rsx_kernels.hpp.)
//
// The kernel is a streaming read with one LDS atomic per key BYTE, so what decides its speed is what a wave-wide
// ds_add costs.  gfx950 serves a 4-byte DS operation in two groups of 32 lanes, {0-31} and {32-63}, over 32 banks
// (bank = word address mod 32); lanes of one group that hit one bank at different addresses take a cycle each.  With
// digit-indexed counters the address is data, so the only way to keep a group conflict-free is to give its 32 lanes 32
// different banks by construction.  Round 1 kept 16 lane-striped copies of every counter (bank = 16 (digit mod 2) +
// lane mod 16): lanes l and l + 16 of a group collided whenever their digits had the same parity, and the counters showed
// 47 % of the LDS cycles lost to conflicts -- about ten cycles per wave-instruction where the issue cost is four.
//
// Here a counter still has 16 copies (stripe = lane mod 16), but the two half-groups of 16 lanes work on DIFFERENT
// COLUMNS in the same instruction, and the columns are laid out in different bank halves:
//
//     word(column c, digit d, stripe s) = ((c mod H) * ROWS + row(d)) * 32 + (c div H) * 16 + s,      H = columns / 2
//
// Instruction i of a key's `columns` atomics counts column i in the lanes with bit 4 of the lane id clear and column
// (i + H) mod columns in the lanes with it set (they extract it as byte i of the key rotated by H bytes).  The first set
// of lanes then lands in bank half (i div H), the second in the other half, and inside a half the bank is the stripe:
// 32 lanes, 32 banks, whatever the digits are.  Only same-address collisions remain (lanes of one stripe with the same
// digit: 4 lanes per stripe and wave-instruction), and columns whose digit is constant across the wave -- the
// column-skip case of radix_sort.hpp:64-70 -- are counted by one lane into a small plain histogram.
//
// 16-bit counters (two digits per word, CTR16: the 8-byte keys, so that two workgroups fit a CU: 64 KiB each) are
// flushed into registers before any of them can reach 2^16; 32-bit counters (4-byte keys and narrower, 64 / 32 KiB)
// never overflow (a workgroup sees less than 2^32 keys).
#pragma once

namespace rsx {

template <typename KT, int BLOCK_ = 1024, int U_ = 2, bool CTR16_ = (sizeof(KT) == 8)> struct HistCfg {
	static constexpr int WC = sizeof(KT);               // columns
	static constexpr int VEC = 16 / sizeof(KT);         // elements per 16-byte lane load
	static constexpr int BLOCK = BLOCK_;
	static constexpr int U = U_;                        // independent 16-byte loads in flight per lane
	static constexpr bool CTR16 = CTR16_ && WC >= 2;
	static constexpr int H = WC >= 2 ? WC / 2 : 1;      // columns per bank half
	static constexpr int ROWS = CTR16 ? 128 : 256;      // words per (column, stripe)
	static constexpr int WORDS = H * ROWS * 32;         // the striped counters
	static constexpr int BINS = WC * 256;
	static constexpr int BPT = (BINS + BLOCK - 1) / BLOCK;   // bins per thread in the reduce
	// A 16-bit counter is fed by the BLOCK / 16 lanes of its stripe, VEC * U keys each per sweep of the main loop.
	static constexpr u32 SWEEPS_PER_FLUSH = CTR16 ? 65535u / ((BLOCK / 16) * VEC * U) : 0xFFFFFFFFu;
	// workgroups per CU the registers are bounded for: as many as the LDS admits, at most 32 waves
	static constexpr int LDS_BYTES = (WORDS + BINS + 1) * 4;
	static constexpr int OCC_LDS = 163840 / LDS_BYTES, OCC_WAVES = 2048 / BLOCK;
	static constexpr int OCC = OCC_LDS < OCC_WAVES ? OCC_LDS : OCC_WAVES;
	static_assert(BLOCK % 64 == 0 && BLOCK >= 64, "whole waves");
};

template <typename C> struct HistSmem {
	u32 ctr[C::WORDS];
	u32 uni[C::BINS];      // wave-uniform columns, counted by one lane (and the scalar fringe)
	u32 descent;
};

// word index of (column, digit) for a lane whose stripe is `s` (0..15); WC == 1: 32 stripes, s = lane mod 32
template <typename C> __device__ __forceinline__ u32 hist_word(u32 col, u32 d, u32 s)
{
	if constexpr (C::WC == 1)
		return d * 32u + s;
	else
		return ((col % C::H) * C::ROWS + (C::CTR16 ? d >> 1 : d)) * 32u + (col / C::H) * 16u + s;
}

template <typename C> __device__ __forceinline__ void hist_add(u32 *ctr, u32 col, u32 d, u32 s)
{
	atomicAdd(&ctr[hist_word<C>(col, d, s)], C::CTR16 ? 1u << ((d & 1u) * 16u) : 1u);
}

// the workgroup's striped counters -> acc[] (thread t owns bins t, t + BLOCK, ...), counters zeroed again
template <typename C> __device__ __forceinline__ void hist_flush(HistSmem<C> &sm, u32 (&acc)[C::BPT], const u32 tid)
{
	__syncthreads();
#pragma unroll
	for (int b = 0; b < C::BPT; ++b) {
		const u32 i = tid + b * C::BLOCK;
		if (i < (u32)C::BINS) {
			const u32 col = i >> 8, d = i & 255u;
			u32 s = 0;
			if constexpr (C::WC == 1) {
#pragma unroll
				for (int r = 0; r < 32; ++r)
					s += sm.ctr[hist_word<C>(col, d, r)];
			} else {
#pragma unroll
				for (int r = 0; r < 16; ++r) {
					const u32 w = sm.ctr[hist_word<C>(col, d, r)];
					s += C::CTR16 ? (w >> ((d & 1u) * 16u)) & 0xFFFFu : w;
				}
			}
			acc[b] += s;
		}
	}
	__syncthreads();
	for (u32 i = tid; i < (u32)C::WORDS; i += C::BLOCK)
		sm.ctr[i] = 0;
	__syncthreads();
}

// grid = blocks_per_seg workgroups (the multi-segment form of round 1 is gone: the sort always used one segment).
// colmask: the columns to count (the MSD split of the multi-GPU path wants one).
// partial: [workgroup][WC * 256] u32 rows for rsx_hist_reduce_kernel; `direct` (few workgroups): the counts are added
// to the histogram at once and no reduce launch follows.
template <typename KT, typename C = HistCfg<KT>>
__global__ __launch_bounds__(C::BLOCK, (C::OCC * C::BLOCK + 255) / 256) void rsx_hist_kernel(const KT *__restrict__ src, u64 n, u32 *__restrict__ partial,
                                                            u32 *__restrict__ unsorted, KdfArgs<KT> ka, u32 colmask = ~0u,
                                                            u64 *__restrict__ direct = nullptr)
{
	constexpr int WC = C::WC, VEC = C::VEC, U = C::U, H = C::H;
	__shared__ HistSmem<C> sm;
	const u32 tid = threadIdx.x;
	const u32 lane = tid & 63;
	const u32 nblk = gridDim.x, blk = blockIdx.x;
	for (u32 i = tid; i < (u32)C::WORDS; i += C::BLOCK)
		sm.ctr[i] = 0;
	for (u32 i = tid; i < (u32)C::BINS; i += C::BLOCK)
		sm.uni[i] = 0;
	if (tid == 0)
		sm.descent = 0;
	__syncthreads();
	colmask &= (1u << WC) - 1u;
	u32 acc[C::BPT];
#pragma unroll
	for (int b = 0; b < C::BPT; ++b)
		acc[b] = 0;

	// elements before the first 16-byte boundary and after the last full vector
	u64 head = ((16 - ((uintptr_t)src & 15)) & 15) / sizeof(KT);
	if (head > n)
		head = n;
	const u64 nvec = (n - head) / VEC;
	const u64 tail_begin = head + nvec * VEC;
	bool descent = false;

	if (blk == 0) {
		// scalar fringe (< 2 * VEC elements)
		const u64 ntail = n - tail_begin;
		for (u64 i = tid; i < head + ntail; i += C::BLOCK) {
			const u64 e = i < head ? i : tail_begin + (i - head);
			const KT k = kdf_apply(src[e], ka);
			if (e + 1 < n && k > kdf_apply(src[e + 1], ka))
				descent = true;
#pragma unroll
			for (int j = 0; j < WC; ++j)
				if ((colmask >> j) & 1u)
					atomicAdd(&sm.uni[j * 256 + ((u32)(k >> (8 * j)) & 0xFFu)], 1u);
		}
	}

	// this lane's half (bit 4 of the lane id) decides which column it counts in instruction i, see the header
	const u32 hb = WC >= 2 ? (lane >> 4) & 1u : 0u;
	const u32 stripe = WC >= 2 ? (lane & 15u) : (lane & 31u);
	u32 lane_off[2];                               // word offset of this lane inside a row of 32, by (i div H)
	lane_off[0] = (hb ? 16u : 0u) + stripe;
	lane_off[1] = (hb ? 0u : 16u) + stripe;
	if constexpr (WC == 1)
		lane_off[0] = lane_off[1] = stripe;

	typedef KT vec_t __attribute__((ext_vector_type(VEC)));
	const vec_t *vsrc = (const vec_t *)(src + head);
	const u64 stride = (u64)nblk * (C::BLOCK * U);
	u32 sweeps = 0;
	// (the loop bound is uniform per workgroup: every wave runs the same number of sweeps, so that the barriers of a
	// flush are reached by all of them; lanes beyond nvec do nothing)
	for (u64 vb = (u64)blk * (C::BLOCK * U); vb < nvec; vb += stride) {
		const u64 v0 = vb + tid;
		vec_t raw[U];
#pragma unroll
		for (int u = 0; u < U; ++u) {
			const u64 v = v0 + (u64)u * C::BLOCK;
			if (v < nvec)
				raw[u] = vsrc[v];
		}
#pragma unroll
		for (int u = 0; u < U; ++u) {
			const u64 v = v0 + (u64)u * C::BLOCK;
			if (v >= nvec)
				break;
			KT k[VEC];
#pragma unroll
			for (int e = 0; e < VEC; ++e)
				k[e] = kdf_apply((KT)raw[u][e], ka);

			// pre-sorted test (radix_sort.hpp:51-54): inside the vector, then against the
			// next element, which the next lane holds except at the wave's right edge.
#pragma unroll
			for (int e = 0; e + 1 < VEC; ++e)
				descent |= k[e] > k[e + 1];
			KT nxt;
			if (sizeof(KT) == 8) {
				const u32 lo = __shfl_down((u32)k[0], 1), hi = __shfl_down((u32)((u64)k[0] >> 32), 1);
				nxt = (KT)(((u64)hi << 32) | lo);
			} else {
				nxt = (KT)__shfl_down((u32)k[0], 1);
			}
			const u64 next_elem = head + (v + 1) * VEC;
			const bool edge = lane == 63 || v + 1 >= nvec;  // the next lane is idle or holds another row
			if (edge)
				nxt = next_elem < n ? kdf_apply(src[next_elem], ka) : k[VEC - 1];
			descent |= k[VEC - 1] > nxt;

			// columns whose digit is identical across the whole wave (the column-skip case,
			// radix_sort.hpp:64-70) are counted by one lane
			const KT first = (KT)(sizeof(KT) == 8
			                          ? (((u64)__builtin_amdgcn_readfirstlane((u32)((u64)k[0] >> 32)) << 32) |
			                             __builtin_amdgcn_readfirstlane((u32)k[0]))
			                          : __builtin_amdgcn_readfirstlane((u32)k[0]));
			KT diff = 0;
#pragma unroll
			for (int e = 0; e < VEC; ++e)
				diff |= (KT)(k[e] ^ first);
			u32 varying = 0;                       // (wave-uniform value: built from ballots)
#pragma unroll
			for (int j = 0; j < WC; ++j)
				if (__any(((u32)(diff >> (8 * j)) & 0xFFu) != 0))
					varying |= 1u << j;
			const u32 cm = colmask & varying;      // columns counted through the striped counters
			const u32 um = colmask & ~varying;     // columns counted by one lane
			if (um) {
				const u64 active = __ballot(1);
				if (mbcnt64(active) == 0) {
#pragma unroll
					for (int j = 0; j < WC; ++j)
						if ((um >> j) & 1u)
							atomicAdd(&sm.uni[j * 256 + ((u32)(first >> (8 * j)) & 0xFFu)], (u32)(VEC * __popcll(active)));
				}
			}
			if (cm) {
				if constexpr (WC == 1) {
#pragma unroll
					for (int e = 0; e < VEC; ++e)
						atomicAdd(&sm.ctr[(u32)k[e] * 32u + stripe], 1u);
				} else {
					// the other half of the lanes sees the key rotated by H bytes and the column mask rotated by H bits
					const u32 cmr = ((cm >> H) | (cm << H)) & ((1u << WC) - 1u);
					const u32 lm = hb ? cmr : cm;
					KT kr[VEC];
#pragma unroll
					for (int e = 0; e < VEC; ++e)
						kr[e] = hb ? (KT)((k[e] >> (8 * H)) | (k[e] << (8 * H))) : k[e];
#pragma unroll
					for (int i = 0; i < WC; ++i) {
						if (!(((cm | cmr) >> i) & 1u))
							continue;              // (uniform: neither half counts anything in this instruction)
						if ((lm >> i) & 1u) {
							u32 *row = sm.ctr + (i % H) * (C::ROWS * 32) + lane_off[i / H];
#pragma unroll
							for (int e = 0; e < VEC; ++e) {
								const u32 d = (u32)(kr[e] >> (8 * i)) & 0xFFu;
								if constexpr (C::CTR16)
									atomicAdd(&row[(d >> 1) * 32u], 1u << ((d & 1u) * 16u));
								else
									atomicAdd(&row[d * 32u], 1u);
							}
						}
					}
				}
			}
		}
		if constexpr (C::CTR16) {
			if (++sweeps == C::SWEEPS_PER_FLUSH) {
				hist_flush<C>(sm, acc, tid);
				sweeps = 0;
			}
		}
	}

	// One flag for the whole array: on unsorted input every wave has seen a descent, and 8192 atomics on one address
	// serialise to about 80 us however small n is.  So: one vote per workgroup, and only while the flag is still clear.
	if (__any(descent) && mbcnt64(__ballot(1)) == 0)
		sm.descent = 1;
	hist_flush<C>(sm, acc, tid);
	if (tid == 0 && sm.descent && __hip_atomic_load(unsorted, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0)
		atomicOr(unsorted, 1u);
	// The workgroup's counts go to its own row of `partial` (plain stores; rsx_hist_reduce_kernel adds the rows up), or
	// straight into the histogram (`direct`).
	u32 *row = partial + (u64)blk * C::BINS;
#pragma unroll
	for (int b = 0; b < C::BPT; ++b) {
		const u32 i = tid + b * C::BLOCK;
		if (i < (u32)C::BINS) {
			const u32 s = acc[b] + sm.uni[i];
			if (direct) {
				if (s)
					atomicAdd(&direct[i], (u64)s);
			} else {
				row[i] = s;
			}
		}
	}
}

// counts[i] += sum over the workgroups of partial[row][i].  grid = (cols256 / 256, HIST_REDUCE_SPLIT): blockIdx.y takes
// every HIST_REDUCE_SPLIT-th row, all its loads in flight at once, and adds its share with one global atomic per bin (32
// per address instead of one per histogram workgroup); `ghist` is zeroed by the caller.
constexpr u32 HIST_REDUCE_SPLIT = 32;
__global__ __launch_bounds__(256) void rsx_hist_reduce_kernel(const u32 *__restrict__ partial, u64 *__restrict__ ghist,
                                                              u32 blocks, u32 cols256)
{
	const u32 i = blockIdx.x * 256 + threadIdx.x;
	const u32 *p = partial + i;
	u64 s = 0;
#pragma unroll 16
	for (u32 b = blockIdx.y; b < blocks; b += HIST_REDUCE_SPLIT)
		s += p[(u64)b * cols256];
	if (s)
		atomicAdd(&ghist[i], s);
}

}  // namespace rsx
