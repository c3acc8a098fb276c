// rsx_hist.hpp -- loop 1 of rs_sort_main (radix_sort.hpp:47-58) on gfx950: every 8-bit column's histogram in ONE read
// of the keys + the pre-sorted test.  (Included by rsx_kernels.hpp.)
//
// The kernel is a streaming read with one LDS atomic per key BYTE.  Measured on MI355X (tools/ubench/hist_probe.hip, 2^28
// keys): a bare read of the same bytes with the same launch shape takes 0.172 ms (u32) / 0.345 ms (u64), i.e. 6.2 TB/s,
// and at that rate a SIMD has about 400 cycles per 16-byte lane vector -- round 1's kernel (0.24 / 0.82 ms) and a first
// conflict-free variant of it (0.27 / 0.70 ms) spent them on INSTRUCTIONS: five or six per key byte (extract, scale, add
// the stripe, pack a 16-bit increment, atomic), a per-vector wave-uniformity test of every column (a compare, a ballot
// and a branch each), two cross-lane shuffles through the LDS crossbar, and a dependent scalar load at the wave's edge
// whose latency every vector waited for.  This version spends two instructions per key byte and almost nothing else:
//
//  * One 256-byte row of counters per DIGIT, shared by all columns: word (column * S + stripe) of row d counts digit d of
//    that column for the lanes of that stripe (S = 64 / columns stripes per column).  The byte address of a counter is
//    then  digit << 8 | per-lane constant:  the digit is dropped into byte 1 of an address register that already holds the
//    lane's constant by ONE instruction (v_mov_b32_sdwa, destination byte 1, the rest preserved, source = the key byte),
//    and the ds_add follows.  64 KiB for every key width; two workgroups per CU.
//  * No bank conflicts by construction: a 4-byte DS operation is served in two groups of 32 lanes over 32 banks, so the
//    32 lanes of a group must hit 32 different words mod 32.  With S stripes, 32 / S lane classes q share a stripe; class
//    q counts column (i xor q) in instruction i (the key's bytes are permuted accordingly by one v_perm_b32 per dword
//    with a per-lane selector), and columns i xor q, q = 0 .. 32/S - 1, lie in different bank groups of the row.
//  * Columns whose digit is constant across the wave (the column-skip case of radix_sort.hpp:64-70, e.g. the zero top
//    bytes of small u64 keys) are counted by one lane into a small plain histogram; the test for that runs only for
//    columns that have not been seen to vary yet in this wave (re-armed now and then): on random keys it costs nothing
//    after the first vector.  A 4-column group with such a column falls back to the unpermuted order for its other
//    columns (a few bank conflicts instead of per-lane predicates).
//  * The neighbour for the pre-sorted test (radix_sort.hpp:51-54) comes through a DPP wave shift; the element after the
//    wave's last one is loaded together with the vectors, not after them.
//  * Keys that are their own KDF (unsigned, ascending: HIST_PLAIN) skip the KDF arithmetic.
#pragma once

namespace rsx {

enum { HIST_GENERIC = 0, HIST_PLAIN = 1 };

template <typename KT, int BLOCK_ = 1024, int U_ = 2> struct HistCfg {
	static constexpr int WC = sizeof(KT);               // columns
	static constexpr int VEC = 16 / sizeof(KT);         // elements per 16-byte lane load
	static constexpr int BLOCK = BLOCK_;
	static constexpr int U = U_;                        // independent 16-byte loads in flight per lane
	static constexpr int STRIPES = 64 / WC;             // lane-striped copies of a counter
	static constexpr int NCLASS = STRIPES >= 32 ? 1 : 32 / STRIPES;   // lane classes sharing a stripe inside a 32-lane group
	static constexpr bool PERM = NCLASS > 1;            // u32 (2 classes), u64 (4 classes)
	static constexpr int BINS = WC * 256;
	static constexpr int BPT = (BINS + BLOCK - 1) / BLOCK;   // bins per thread in the reduce
	static constexpr int LDS_BYTES = 65536 + BINS * 4 + 16;
	static constexpr int OCC_LDS = 163840 / LDS_BYTES, OCC_WAVES = 2048 / BLOCK;
	static constexpr int OCC = OCC_LDS < OCC_WAVES ? OCC_LDS : OCC_WAVES;   // workgroups per CU the registers are bounded for
	static constexpr u32 REARM = 64;                    // sweeps after which every column is tested for wave-uniformity again
	static_assert(BLOCK % 64 == 0 && BLOCK >= 64, "whole waves");
};

template <typename C> struct HistSmem {
	u32 ctr[16384];        // [digit][column * STRIPES + stripe]; FIRST member: a counter's byte address is digit << 8 | low byte
	u32 uni[C::BINS];      // wave-uniform columns, counted by one lane (and the scalar fringe)
	u32 descent;
};

// addr.byte1 = src.byte<B>, everything else of addr preserved
template <int B> __device__ __forceinline__ void hist_put_digit(u32 &addr, const u32 src)
{
	if constexpr (B == 0)
		asm("v_mov_b32_sdwa %0, %1 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_0" : "+v"(addr) : "v"(src));
	else if constexpr (B == 1)
		asm("v_mov_b32_sdwa %0, %1 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_1" : "+v"(addr) : "v"(src));
	else if constexpr (B == 2)
		asm("v_mov_b32_sdwa %0, %1 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_2" : "+v"(addr) : "v"(src));
	else
		asm("v_mov_b32_sdwa %0, %1 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_3" : "+v"(addr) : "v"(src));
}

// lane i receives lane i + 1's value (lane 63 keeps its own): a DPP wave shift, no LDS crossbar
__device__ __forceinline__ u32 wave_next_lane(const u32 x)
{
	return (u32)__builtin_amdgcn_update_dpp((int)x, (int)x, 0x130 /* wave_shl:1 */, 0xF, 0xF, false);
}

// the workgroup's striped counters (+ the plain ones) -> acc[] (thread t owns bins t, t + BLOCK, ...)
template <typename C> __device__ __forceinline__ void hist_collect(HistSmem<C> &sm, u32 (&acc)[C::BPT], const u32 tid)
{
#pragma unroll
	for (int b = 0; b < C::BPT; ++b) {
		const u32 i = tid + b * C::BLOCK;
		acc[b] = 0;
		if (i < (u32)C::BINS) {
			const u32 col = i >> 8, d = i & 255u;
			u32 s = sm.uni[i];
#pragma unroll
			for (int r = 0; r < C::STRIPES; ++r)
				s += sm.ctr[d * 64u + col * C::STRIPES + ((r + tid) & (C::STRIPES - 1))];   // (rotated: neighbouring lanes start on different banks)
			acc[b] = s;
		}
	}
}

// FUSED (the blocking sorts): before it counts, every workgroup zeroes its share of three regions -- the status words of THIS
// sort's passes and the flags / histogram of the NEXT sort (the host alternates between two sets: the set this sort counts
// into was zeroed by the previous sort) --, which is one launch less per sort (rsx_zero3_kernel); at 10^5 keys a launch
// costs as much as any of the sort's kernels.
// (Also tried: the plan made by the workgroup that finishes last.  Every workgroup then needs a device-scope release fence
// before it signs off, which on this multi-die part writes the die's L2 back: 2-3 us each, 22 us instead of 5 for the
// histogram kernel of 2^16 keys and 48 instead of 8 for 2^20.  rsx_plan_all_kernel, one workgroup, does it instead.)
struct HistFuse {
	u32x4 *z0, *z1, *z2;
	u64 n0, n1, n2;        // 16-byte units
	// a device-scheduled sort (rsx_sort_inplace_async) that has first tried to do without this kernel: *gate == GATE_DONE
	// (SegCtl::mode == SEG_MODE_LEAVES, rsx_hybrid.hpp) says that the keys are sorted already -- nothing is read or counted
	const u32 *gate;
};
constexpr u32 GATE_DONE = 1;

// colmask: the columns to count (the MSD split of the multi-GPU path wants one).
// partial: [workgroup][WC * 256] u32 rows for rsx_hist_reduce_kernel; `direct` (few workgroups): the counts are added
// to the histogram at once and no reduce launch follows.
template <typename KT, typename C = HistCfg<KT>, int MODE = HIST_GENERIC, bool FUSED = false>
__global__ __launch_bounds__(C::BLOCK, (C::OCC * C::BLOCK + 255) / 256) void rsx_hist_kernel(const KT *__restrict__ src, u64 n,
                                                                                             u32 *__restrict__ partial,
                                                                                             u32 *__restrict__ unsorted, KdfArgs<KT> ka,
                                                                                             u32 colmask = ~0u,
                                                                                             u64 *__restrict__ direct = nullptr,
                                                                                             HistFuse fuse = HistFuse{})
{
	constexpr int WC = C::WC, VEC = C::VEC, U = C::U, S = C::STRIPES;
	constexpr int KD = WC >= 4 ? WC / 4 : 1;           // dwords per key
	__shared__ HistSmem<C> sm;
	const u32 tid = threadIdx.x;
	const u32 lane = tid & 63;
	const u32 nblk = gridDim.x, blk = blockIdx.x;
	if (fuse.gate && *fuse.gate == GATE_DONE)
		return;
	if constexpr (FUSED) {
		const u64 stride = (u64)nblk * C::BLOCK, t = (u64)blk * C::BLOCK + tid;
		const u32x4 z = {0u, 0u, 0u, 0u};
		for (u64 i = t; i < fuse.n0; i += stride)
			fuse.z0[i] = z;
		for (u64 i = t; i < fuse.n1; i += stride)
			fuse.z1[i] = z;
		for (u64 i = t; i < fuse.n2; i += stride)
			fuse.z2[i] = z;
	}
	for (u32 i = tid; i < 16384; i += C::BLOCK)
		sm.ctr[i] = 0;
	for (u32 i = tid; i < (u32)C::BINS; i += C::BLOCK)
		sm.uni[i] = 0;
	if (tid == 0)
		sm.descent = 0;
	__syncthreads();
	colmask &= (1u << WC) - 1u;

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

	// ---- per-lane constants: the stripe, the class q, the byte selector of v_perm_b32 (byte i of the permuted dword = byte
	// i xor q of the key's dword) and the address registers (low byte = 4 * (column * S + stripe); byte 1 receives the digit)
	const u32 stripe = lane & (S - 1);
	const u32 q = C::PERM ? (lane / S) & (C::NCLASS - 1) : 0u;
	const u32 sel = 0x03020100u ^ (q * 0x01010101u);
	u32 aperm[C::PERM ? WC : 1];                        // slot (dword w, byte b) counts column 4 w + (b xor q)
	aperm[0] = 0;
	if constexpr (C::PERM) {
#pragma unroll
		for (int s = 0; s < WC; ++s)
			aperm[s] = (((u32)(s & ~3) + (((u32)s & 3u) ^ q)) * S + stripe) * 4u;
	}
	u32 aid = stripe * 4u;                              // unpermuted order: the column's offset is an immediate
	char *const ctr_bytes = (char *)sm.ctr;

	typedef KT vec_t __attribute__((ext_vector_type(VEC)));
	const vec_t *vsrc = (const vec_t *)(src + head);
	const u64 stride = (u64)nblk * (C::BLOCK * U);
	u32 cand = colmask;                                 // columns not seen to vary yet in this wave (a wave-uniform value)
	u32 sweeps = 0;
	// One sweep: U vectors per lane.  FULL: every lane of the workgroup has all its U vectors (the test is uniform, so
	// everything derived from ballots stays in scalar registers and the branches on it are scalar branches).
	auto sweep = [&](auto full_c, const u64 vb) {
		constexpr bool FULL = decltype(full_c)::value;
		const u64 v0 = vb + tid;
		vec_t raw[U];
		KT edge[U];
#pragma unroll
		for (int u = 0; u < U; ++u) {
			const u64 v = v0 + (u64)u * C::BLOCK;
			if (FULL || v < nvec)
				raw[u] = vsrc[v];
		}
#pragma unroll
		for (int u = 0; u < U; ++u) {
			// the element after this lane's vector, where the next lane does not hold it: the wave's right edge and the
			// end of the array (one extra lane load per wave and vector, issued with the vectors)
			const u64 v = v0 + (u64)u * C::BLOCK;
			const u64 next_elem = head + (v + 1) * VEC;
			edge[u] = 0;
			if ((FULL || v < nvec) && (lane == 63 || v + 1 >= nvec) && next_elem < n)
				edge[u] = src[next_elem];
		}
#pragma unroll
		for (int u = 0; u < U; ++u) {
			const u64 v = v0 + (u64)u * C::BLOCK;
			if (!FULL && v >= nvec)
				break;
			KT k[VEC];
#pragma unroll
			for (int e = 0; e < VEC; ++e)
				k[e] = MODE == HIST_PLAIN ? (KT)raw[u][e] : kdf_apply((KT)raw[u][e], ka);

			// pre-sorted test (radix_sort.hpp:51-54): inside the vector, then against the next element
#pragma unroll
			for (int e = 0; e + 1 < VEC; ++e)
				descent |= k[e] > k[e + 1];
			KT nxt;
			if constexpr (sizeof(KT) == 8)
				nxt = (KT)(((u64)wave_next_lane((u32)((u64)k[0] >> 32)) << 32) | wave_next_lane((u32)k[0]));
			else
				nxt = (KT)wave_next_lane((u32)k[0]);
			const u64 next_elem = head + (v + 1) * VEC;
			if (lane == 63 || v + 1 >= nvec)
				nxt = next_elem < n ? (MODE == HIST_PLAIN ? edge[u] : kdf_apply(edge[u], ka)) : k[VEC - 1];
			descent |= k[VEC - 1] > nxt;

			// columns whose digit is identical across the whole wave are counted by one lane; only columns that have not
			// been seen to vary are tested (a column counted through the striped counters is always counted correctly)
			u32 um = 0;
			if (cand) {
				const KT first = (KT)(sizeof(KT) == 8
				                          ? (((u64)__builtin_amdgcn_readfirstlane((u32)((u64)k[0] >> 32)) << 32) |
				                             __builtin_amdgcn_readfirstlane((u32)k[0]))
				                          : __builtin_amdgcn_readfirstlane((u32)k[0]));
				KT diff = 0;
#pragma unroll
				for (int e = 0; e < VEC; ++e)
					diff |= (KT)(k[e] ^ first);
#pragma unroll
				for (int j = 0; j < WC; ++j)
					if (((cand >> j) & 1u) && __any(((u32)(diff >> (8 * j)) & 0xFFu) != 0))
						cand &= ~(1u << j);
				cand = __builtin_amdgcn_readfirstlane(cand);
				um = cand;
				if (um) {
					const u64 active = __ballot(1);
					if (mbcnt64(active) == 0) {
#pragma unroll
						for (int j = 0; j < WC; ++j)
							if ((um >> j) & 1u)
								atomicAdd(&sm.uni[j * 256 + ((u32)(first >> (8 * j)) & 0xFFu)], (u32)(VEC * __popcll(active)));
					}
				}
			}
			// columns the striped counters do not see in this vector (wave-uniform)
			const u32 skip = __builtin_amdgcn_readfirstlane(um | (~colmask & ((1u << WC) - 1u)));

			if constexpr (WC >= 4) {
#pragma unroll
				for (int e = 0; e < VEC; ++e) {
#pragma unroll
					for (int w = 0; w < KD; ++w) {
						const u32 kd = w == 0 ? (u32)k[e] : (u32)((u64)k[e] >> 32);
						const u32 sg = (skip >> (4 * w)) & 0xFu;
						if (sg == 0) {
							// all four columns of the group: permuted order, no bank conflicts
							const u32 kp = __builtin_amdgcn_perm(kd, kd, sel);
							hist_put_digit<0>(aperm[4 * w + 0], kp);
							atomicAdd((u32 *)(ctr_bytes + aperm[4 * w + 0]), 1u);
							hist_put_digit<1>(aperm[4 * w + 1], kp);
							atomicAdd((u32 *)(ctr_bytes + aperm[4 * w + 1]), 1u);
							hist_put_digit<2>(aperm[4 * w + 2], kp);
							atomicAdd((u32 *)(ctr_bytes + aperm[4 * w + 2]), 1u);
							hist_put_digit<3>(aperm[4 * w + 3], kp);
							atomicAdd((u32 *)(ctr_bytes + aperm[4 * w + 3]), 1u);
						} else if (sg != 0xFu) {
							// some columns of the group are skipped: the others in key order
							if (!(sg & 1u)) {
								hist_put_digit<0>(aid, kd);
								atomicAdd((u32 *)(ctr_bytes + aid + (4 * w + 0) * S * 4), 1u);
							}
							if (!(sg & 2u)) {
								hist_put_digit<1>(aid, kd);
								atomicAdd((u32 *)(ctr_bytes + aid + (4 * w + 1) * S * 4), 1u);
							}
							if (!(sg & 4u)) {
								hist_put_digit<2>(aid, kd);
								atomicAdd((u32 *)(ctr_bytes + aid + (4 * w + 2) * S * 4), 1u);
							}
							if (!(sg & 8u)) {
								hist_put_digit<3>(aid, kd);
								atomicAdd((u32 *)(ctr_bytes + aid + (4 * w + 3) * S * 4), 1u);
							}
						}
					}
				}
			} else {
				// 1- and 2-byte keys: 64 or 32 stripes, conflict-free in key order
#pragma unroll
				for (int e = 0; e < VEC; ++e) {
					const u32 kd = (u32)k[e];
					if (!(skip & 1u)) {
						hist_put_digit<0>(aid, kd);
						atomicAdd((u32 *)(ctr_bytes + aid), 1u);
					}
					if (WC == 2 && !(skip & 2u)) {
						hist_put_digit<1>(aid, kd);
						atomicAdd((u32 *)(ctr_bytes + aid + S * 4), 1u);
					}
				}
			}
		}
	};
	for (u64 vb = (u64)blk * (C::BLOCK * U); vb < nvec; vb += stride) {
		if (vb + (u64)C::BLOCK * U < nvec)      // (strictly: the last vector's right neighbour is then a vector too)
			sweep(std::true_type{}, vb);
		else
			sweep(std::false_type{}, vb);
		if (++sweeps == C::REARM) {
			sweeps = 0;
			cand = colmask;
		}
	}

	// One flag for the whole array: on unsorted input every wave has seen a descent, and 8192 atomics on one address
	// serialise to about 80 us however small n is.  So: one vote per workgroup, and only while the flag is still clear.
	if (__any(descent) && mbcnt64(__ballot(1)) == 0)
		sm.descent = 1;
	__syncthreads();
	if (tid == 0 && sm.descent && __hip_atomic_load(unsorted, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0)
		atomicOr(unsorted, 1u);
	// The workgroup's counts go to its own row of `partial` (plain stores; rsx_hist_reduce_kernel adds the rows up), or
	// straight into the histogram (`direct`).
	u32 acc[C::BPT];
	hist_collect<C>(sm, acc, tid);
	u32 *row = partial + (u64)blk * C::BINS;
#pragma unroll
	for (int b = 0; b < C::BPT; ++b) {
		const u32 i = tid + b * C::BLOCK;
		if (i < (u32)C::BINS) {
			if (direct) {
				if (acc[b])
					atomicAdd(&direct[i], (u64)acc[b]);
			} else {
				row[i] = acc[b];
			}
		}
	}
}

// counts[i] += sum over the workgroups of partial[row][i].  grid = (cols256 / 256, HIST_REDUCE_SPLIT): blockIdx.y takes
// every HIST_REDUCE_SPLIT-th row, all its loads in flight at once, and adds its share with one global atomic per bin (32
// per address instead of one per histogram workgroup); `ghist` is zeroed by the caller.
constexpr u32 HIST_REDUCE_SPLIT = 32;
__global__ __launch_bounds__(256) void rsx_hist_reduce_kernel(const u32 *__restrict__ partial, u64 *__restrict__ ghist,
                                                              u32 blocks, u32 cols256, const u32 *gate = nullptr)
{
	if (gate && *gate == GATE_DONE)
		return;
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
