// rsx_small.hpp -- the whole of rs_sort_main (radix_sort.hpp:31-93) in ONE workgroup, for arrays that fit LDS twice
// (n * sizeof(key) <= 64 KiB: 16 Ki four-byte keys): histogram of all columns + pre-sorted test, column probe,
// one stable LDS-to-LDS scatter per kept column, result written to `src` or `aux` by the parity rule.  One launch
// instead of a dozen: the general path costs about 75 us however small n is (histogram, reduce, plan x 2, memsets and
// a scatter kernel per column), this one about 15 us.
//
// Ranking is the same as in rsx_scatter2_kernel: a wave owns a contiguous slice, counts its digits into its row of
// (wave, digit) cells, the cells become run starts by a prefix over waves and digits, and a returning LDS atomic on the
// cell is the key's position.  It rests on returning LDS atomics resolving same-address lanes in lane order, which the
// host verifies on the device before it uses this kernel (lds_order_selfcheck, rsx.hip).
#pragma once

#include "rsx_kernels.hpp"

namespace rsx {

constexpr u32 SMALL_SORT_BYTES = 65536;   // per LDS buffer

template <typename KT>
__global__ __launch_bounds__(1024) void rsx_small_sort_kernel(KT *__restrict__ src, KT *__restrict__ aux, u32 n, KdfArgs<KT> ka,
                                                              Plan *__restrict__ plan_out,   // pinned host memory
                                                              bool inplace = false)          // result in src whatever the parity
{
	constexpr int WC = sizeof(KT), NW = 16, BLOCK = 1024;
	constexpr u32 CAP = SMALL_SORT_BYTES / sizeof(KT);
	__shared__ KT buf[2][CAP];
	__shared__ u32 cell[NW][256];
	__shared__ u32 hist[WC][256];
	__shared__ u32 wsum[4];
	__shared__ u32 s_unsorted, s_ncols, s_cols[8];
	const u32 tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;

	for (u32 i = tid; i < WC * 256; i += BLOCK)
		(&hist[0][0])[i] = 0;
	if (tid == 0)
		s_unsorted = 0;
	for (u32 i = tid; i < n; i += BLOCK)
		buf[0][i] = src[i];
	__syncthreads();

	// radix_sort.hpp:47-58: every column's histogram and the ordered-neighbour test
	bool descent = false;
	for (u32 i = tid; i < n; i += BLOCK) {
		const KT k = kdf_apply(buf[0][i], ka);
		if (i + 1 < n && k > kdf_apply(buf[0][i + 1], ka))
			descent = true;
#pragma unroll
		for (int j = 0; j < WC; ++j)
			atomicAdd(&hist[j][(u32)(k >> (8 * j)) & 0xFFu], 1u);
	}
	if (__any(descent) && mbcnt64(__ballot(1)) == 0)
		s_unsorted = 1;
	__syncthreads();

	// :60-70: early exit, column probe on the first key
	if (tid == 0) {
		const KT k0 = kdf_apply(buf[0][0], ka);
		u32 nc = 0;
		for (int j = 0; j < WC; ++j)
			if (hist[j][(u32)(k0 >> (8 * j)) & 0xFFu] != n)
				s_cols[nc++] = (u32)j;
		s_ncols = nc;
		plan_out->sorted = s_unsorted ? 0u : 1u;
		plan_out->ncols = s_unsorted ? nc : 0u;
		for (u32 j = 0; j < 8; ++j)
			plan_out->cols[j] = j < nc ? s_cols[j] : 0u;
	}
	__syncthreads();
	if (!s_unsorted)
		return;
	const u32 ncols = s_ncols;

	// :82-90: one stable scatter per kept column, LDS to LDS.  Wave w owns [w * per, (w + 1) * per) of the array
	// (per a multiple of 64); round r of a lane is element w * per + 64 r + lane.
	const u32 per = ((n + NW * 64 - 1) / (NW * 64)) * 64;
	const u32 wbeg = wid * per, wend = wbeg + per < n ? wbeg + per : n;
	u32 cur = 0;
	for (u32 c = 0; c < ncols; ++c) {
		const u32 shift = 8 * s_cols[c];
		const KT *in = buf[cur];
		KT *out = buf[cur ^ 1];
#pragma unroll
		for (int k = 0; k < 4; ++k)
			cell[wid][lane + 64 * k] = 0;
		// (a wave's DS operations execute in order: no barrier between zeroing and counting its own row)
		for (u32 i = wbeg + lane; i < wend; i += 64)
			atomicAdd(&cell[wid][(u32)(kdf_apply(in[i], ka) >> shift) & 0xFFu], 1u);
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
			u32 acc = incl - tot;   // :72-80, the exclusive scan over the digits
			for (u32 w = 0; w < wid; ++w)
				acc += wsum[w];
#pragma unroll
			for (int w = 0; w < NW; ++w) {
				const u32 cnt = cell[w][tid];
				cell[w][tid] = acc;
				acc += cnt;
			}
		}
		__syncthreads();
		for (u32 i = wbeg + lane; i < wend; i += 64) {
			const KT key = in[i];
			const u32 pos = __hip_atomic_fetch_add(&cell[wid][(u32)(kdf_apply(key, ka) >> shift) & 0xFFu], 1u, __ATOMIC_RELAXED,
			                                       __HIP_MEMORY_SCOPE_WORKGROUP);
			out[pos] = key;
		}
		__syncthreads();
		cur ^= 1;
	}
	KT *dst = ((ncols & 1) && !inplace) ? aux : src;   // :92
	for (u32 i = tid; i < n; i += BLOCK)
		dst[i] = buf[cur][i];
}

}  // namespace rsx

namespace rsx {

// The same for keys with a payload (RANK = false: rsx_sort_pairs_device) and for the stable argsort (RANK = true:
// radix_sort_rank.hpp:22-92 with Listing 6's semantics; the payload is the element index, the keys are not written back):
// 2 x (keys + payloads) of the array must fit SMALL_PAIR_BYTES of LDS.
//   pairs: result in (k0, v0) for an even number of kept columns, in (k1, v1) for an odd one; sorted input untouched.
//   rank:  v0 = index_buffer, v1 = index_buffer + n; sorted input: v0 = 0 .. n-1, v1 untouched (:52,:55-57).
constexpr u32 SMALL_PAIR_BYTES = 131072;

template <typename KT, typename VT, bool RANK>
__global__ __launch_bounds__(1024) void rsx_small_pairs_kernel(const KT *__restrict__ k0, KT *__restrict__ k1, VT *__restrict__ v0,
                                                               VT *__restrict__ v1, u32 n, KdfArgs<KT> ka,
                                                               Plan *__restrict__ plan_out,
                                                               bool inplace = false)   // result in (k0, v0) whatever the parity
{
	constexpr int WC = sizeof(KT), NW = 16, BLOCK = 1024;
	constexpr u32 CAP = SMALL_PAIR_BYTES / (2 * (sizeof(KT) + sizeof(VT)));
	__shared__ KT kbuf[2][CAP];
	__shared__ VT vbuf[2][CAP];
	__shared__ u32 cell[NW][256];
	__shared__ u32 hist[WC][256];
	__shared__ u32 wsum[4];
	__shared__ u32 s_unsorted, s_ncols, s_cols[8];
	const u32 tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;

	for (u32 i = tid; i < WC * 256; i += BLOCK)
		(&hist[0][0])[i] = 0;
	if (tid == 0)
		s_unsorted = 0;
	for (u32 i = tid; i < n; i += BLOCK) {
		kbuf[0][i] = k0[i];
		vbuf[0][i] = RANK ? (VT)i : v0[i];
	}
	__syncthreads();
	bool descent = false;
	for (u32 i = tid; i < n; i += BLOCK) {
		const KT k = kdf_apply(kbuf[0][i], ka);
		if (i + 1 < n && k > kdf_apply(kbuf[0][i + 1], ka))
			descent = true;
#pragma unroll
		for (int j = 0; j < WC; ++j)
			atomicAdd(&hist[j][(u32)(k >> (8 * j)) & 0xFFu], 1u);
	}
	if (__any(descent) && mbcnt64(__ballot(1)) == 0)
		s_unsorted = 1;
	__syncthreads();
	if (tid == 0) {
		const KT key0 = kdf_apply(kbuf[0][0], ka);
		u32 nc = 0;
		for (int j = 0; j < WC; ++j)
			if (hist[j][(u32)(key0 >> (8 * j)) & 0xFFu] != n)
				s_cols[nc++] = (u32)j;
		s_ncols = nc;
		plan_out->sorted = s_unsorted ? 0u : 1u;
		plan_out->ncols = s_unsorted ? nc : 0u;
		for (u32 j = 0; j < 8; ++j)
			plan_out->cols[j] = j < nc ? s_cols[j] : 0u;
	}
	__syncthreads();
	if (!s_unsorted) {
		if (RANK)
			for (u32 i = tid; i < n; i += BLOCK)
				v0[i] = (VT)i;
		return;
	}
	const u32 ncols = s_ncols;
	const u32 per = ((n + NW * 64 - 1) / (NW * 64)) * 64;
	const u32 wbeg = wid * per, wend = wbeg + per < n ? wbeg + per : n;
	u32 cur = 0;
	for (u32 c = 0; c < ncols; ++c) {
		const u32 shift = 8 * s_cols[c];
		const KT *kin = kbuf[cur];
		const VT *vin = vbuf[cur];
		KT *kout = kbuf[cur ^ 1];
		VT *vout = vbuf[cur ^ 1];
#pragma unroll
		for (int k = 0; k < 4; ++k)
			cell[wid][lane + 64 * k] = 0;
		for (u32 i = wbeg + lane; i < wend; i += 64)
			atomicAdd(&cell[wid][(u32)(kdf_apply(kin[i], ka) >> shift) & 0xFFu], 1u);
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
				const u32 cnt = cell[w][tid];
				cell[w][tid] = acc;
				acc += cnt;
			}
		}
		__syncthreads();
		for (u32 i = wbeg + lane; i < wend; i += 64) {
			const KT key = kin[i];
			const u32 pos = __hip_atomic_fetch_add(&cell[wid][(u32)(kdf_apply(key, ka) >> shift) & 0xFFu], 1u, __ATOMIC_RELAXED,
			                                       __HIP_MEMORY_SCOPE_WORKGROUP);
			kout[pos] = key;
			vout[pos] = vin[i];
		}
		__syncthreads();
		cur ^= 1;
	}
	VT *vdst = ((ncols & 1) && !inplace) ? v1 : v0;   // radix_sort.hpp:92 / radix_sort_rank.hpp:91
	for (u32 i = tid; i < n; i += BLOCK)
		vdst[i] = vbuf[cur][i];
	if (!RANK) {
		KT *kdst = ((ncols & 1) && !inplace) ? k1 : const_cast<KT *>(k0);
		for (u32 i = tid; i < n; i += BLOCK)
			kdst[i] = kbuf[cur][i];
	}
}

}  // namespace rsx
