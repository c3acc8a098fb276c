// rsx_hist_r1.hpp -- round 1's histogram kernel (16 lane-striped copies per bin, bank = 16 (digit mod 2) + lane mod 16),
// kept for tools/ubench/hist_probe.hip to measure the present kernel (radix_sorting_amd/csrc/rsx_hist.hpp) against.
#pragma once
#include "rsx_kernels.hpp"
namespace rsx {
// =============================================================================
// Segments (histogram / plan kernels)
// =============================================================================
// The histogram can be kept per contiguous segment of the input: counts[seg][column][256] (u64),
// turned by the plan kernel into the exclusive offset of (segment, digit).  The sort itself runs
// with one segment (per-segment counts of the input order are only valid for the first pass);
// the MSD split of the multi-GPU path uses the same kernels.

// =============================================================================
// Kernel 1: histogram of all columns + pre-sorted test
// =============================================================================

// Measured on MI355X, 2^28 u32 (tools/ubench/hist_probe.hip): the kernel is bound by the LDS atomics (four per key,
// about ten cycles per wave-instruction and CU), so what matters is a full CU (32 waves = two workgroups of 1024), few
// workgroups, and no more than two or three 16-byte loads in flight per lane (0.23-0.27 ms; four: 0.32; 2048
// workgroups of 256: 0.36).
template <typename KT, int BLOCK_ = 1024, int U_ = 2, int R_ = (sizeof(KT) == 8 ? 8 : 16)> struct HistR1Cfg {
	static constexpr int WC = sizeof(KT);               // columns
	static constexpr int VEC = 16 / sizeof(KT);         // elements per 16-byte lane load
	static constexpr int R = R_;                        // lane-striped copies per bin
	static constexpr int BLOCK = BLOCK_;
	static constexpr int U = U_;                        // independent 16-byte loads in flight per lane
};

template <typename KT, int R>
__device__ __forceinline__ void hist_r1_add_one(u32 *lh, KT k, u32 lane, u32 colmask)
{
#pragma unroll
	for (int j = 0; j < (int)sizeof(KT); ++j) {
		if (!((colmask >> j) & 1u))
			continue;
		const u32 d = (u32)(k >> (8 * j)) & 0xFFu;
		atomicAdd(&lh[(j * 256 + d) * R + (lane & (R - 1))], 1u);
	}
}

// grid = nseg * blocks_per_seg.  With nseg > 1 the host guarantees that src is 16-byte aligned
// and seg_elems is a multiple of VEC.
template <typename KT, typename C = HistR1Cfg<KT>>
__global__ __launch_bounds__(C::BLOCK) void rsx_hist_r1_kernel(const KT *__restrict__ src, u64 n, u32 *__restrict__ partial,
                                                            u32 *__restrict__ unsorted, KdfArgs<KT> ka, u32 nseg,
                                                            u32 blocks_per_seg, u64 seg_elems, u32 colmask = ~0u,
                                                            u64 *__restrict__ direct = nullptr)
{
	// colmask: the columns to count (the MSD split of the multi-GPU path wants one: a quarter of the LDS atomics)
	constexpr int WC = C::WC, VEC = C::VEC, R = C::R, U = C::U;
	__shared__ u32 lh[WC * 256 * R];
	__shared__ u32 s_descent;
	const u32 tid = threadIdx.x;
	const u32 lane = tid & 63;
	const u32 seg = blockIdx.x / blocks_per_seg, bis = blockIdx.x % blocks_per_seg;
	for (u32 i = tid; i < WC * 256 * R; i += C::BLOCK)
		lh[i] = 0;
	if (tid == 0)
		s_descent = 0;
	__syncthreads();

	// elements before the first 16-byte boundary (single-segment launches only) and after the last full vector
	u64 head = nseg > 1 ? 0 : ((16 - ((uintptr_t)src & 15)) & 15) / sizeof(KT);
	if (head > n)
		head = n;
	const u64 nvec = (n - head) / VEC;
	const u64 tail_begin = head + nvec * VEC;
	const u64 vbeg = nseg > 1 ? (u64)seg * (seg_elems / VEC) : 0;
	u64 vend = nseg > 1 ? vbeg + seg_elems / VEC : nvec;
	if (vend > nvec)
		vend = nvec;
	bool descent = false;

	if (bis == 0 && (seg == 0 || seg == nseg - 1)) {
		// scalar fringe (< 2*VEC elements): the head belongs to segment 0, the tail to the last segment
		const u64 nhead = seg == 0 ? head : 0;
		const u64 ntail = seg == nseg - 1 ? n - tail_begin : 0;
		for (u64 i = tid; i < nhead + ntail; i += C::BLOCK) {
			const u64 e = i < nhead ? i : tail_begin + (i - nhead);
			const KT k = kdf_apply(src[e], ka);
			if (e + 1 < n && k > kdf_apply(src[e + 1], ka))
				descent = true;
			hist_r1_add_one<KT, R>(lh, k, lane, colmask);
		}
	}

	typedef KT vec_t __attribute__((ext_vector_type(VEC)));
	const vec_t *vsrc = (const vec_t *)(src + head);
	const u64 stride = (u64)blocks_per_seg * (C::BLOCK * U);
	for (u64 v0 = vbeg + (u64)bis * (C::BLOCK * U) + tid; v0 < vend; v0 += stride) {
		vec_t raw[U];
#pragma unroll
		for (int u = 0; u < U; ++u) {
			const u64 v = v0 + (u64)u * C::BLOCK;
			if (v < vend)
				raw[u] = vsrc[v];
		}
#pragma unroll
		for (int u = 0; u < U; ++u) {
			const u64 v = v0 + (u64)u * C::BLOCK;
			if (v >= vend)
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
			const bool edge = lane == 63 || v + 1 >= vend;  // the next lane is idle or holds another row
			if (edge)
				nxt = next_elem < n ? kdf_apply(src[next_elem], ka) : k[VEC - 1];
			descent |= k[VEC - 1] > nxt;

			// histogram.  A column whose digit is identical across the whole wave (the
			// column-skip case, radix_sort.hpp:64-70) is counted by one lane.
			const KT first = (KT)(sizeof(KT) == 8
			                          ? (((u64)__builtin_amdgcn_readfirstlane((u32)((u64)k[0] >> 32)) << 32) |
			                             __builtin_amdgcn_readfirstlane((u32)k[0]))
			                          : __builtin_amdgcn_readfirstlane((u32)k[0]));
			KT diff = 0;
#pragma unroll
			for (int e = 0; e < VEC; ++e)
				diff |= (KT)(k[e] ^ first);
			const u64 active = __ballot(1);
#pragma unroll
			for (int j = 0; j < WC; ++j) {
				if (!((colmask >> j) & 1u))
					continue;
				const bool differs = ((u32)(diff >> (8 * j)) & 0xFFu) != 0;
				if (__any(differs)) {
#pragma unroll
					for (int e = 0; e < VEC; ++e) {
						const u32 d = (u32)(k[e] >> (8 * j)) & 0xFFu;
						atomicAdd(&lh[(j * 256 + d) * R + (lane & (R - 1))], 1u);
					}
				} else if (mbcnt64(active) == 0) {
					const u32 d = (u32)(first >> (8 * j)) & 0xFFu;
					atomicAdd(&lh[(j * 256 + d) * R], (u32)(VEC * __popcll(active)));
				}
			}
		}
	}

	// One flag for the whole array: on unsorted input every wave has seen a descent, and 8192 atomics on one address
	// serialise to about 80 us however small n is.  So: one vote per workgroup, and only while the flag is still clear.
	if (__any(descent) && mbcnt64(__ballot(1)) == 0)
		s_descent = 1;

	// The workgroup's counts go to its own row of `partial` (plain stores; rsx_hist_r1_reduce_kernel adds the rows up).
	__syncthreads();
	if (tid == 0 && s_descent && __hip_atomic_load(unsorted, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0)
		atomicOr(unsorted, 1u);
	// `direct` (few workgroups): the counts are added to the segment's histogram at once, no reduce launch follows
	u32 *row = partial + (u64)blockIdx.x * (WC * 256);
	for (u32 i = tid; i < WC * 256; i += C::BLOCK) {
		u32 s = 0;
#pragma unroll
		for (int r = 0; r < R; ++r)
			s += lh[i * R + r];
		if (direct) {
			if (s)
				atomicAdd(&direct[(u64)seg * (WC * 256) + i], (u64)s);
		} else {
			row[i] = s;
		}
	}
}

// counts[seg][i] += sum over the segment's workgroups of partial[row][i].  grid = (nseg * cols256 / 256, HIST_R1_REDUCE_SPLIT):
// blockIdx.y takes every HIST_R1_REDUCE_SPLIT-th row, all its loads in flight at once, and adds its share with one global
// atomic per bin (32 per address instead of one per histogram workgroup); `ghist` is zeroed by the caller.
constexpr u32 HIST_R1_REDUCE_SPLIT = 32;
__global__ __launch_bounds__(256) void rsx_hist_r1_reduce_kernel(const u32 *__restrict__ partial, u64 *__restrict__ ghist,
                                                              u32 blocks_per_seg, u32 cols256)
{
	const u32 per_seg = cols256 / 256, seg = blockIdx.x / per_seg, i = (blockIdx.x % per_seg) * 256 + threadIdx.x;
	const u32 *p = partial + (u64)seg * blocks_per_seg * cols256 + i;
	u64 s = 0;
#pragma unroll 16
	for (u32 b = blockIdx.y; b < blocks_per_seg; b += HIST_R1_REDUCE_SPLIT)
		s += p[(u64)b * cols256];
	if (s)
		atomicAdd(&ghist[(u64)seg * cols256 + i], s);
}

}  // namespace rsx
