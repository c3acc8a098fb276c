// EXPERIMENT (round 2), not part of the library.  Measured on 2^28 u32 keys (profiles/r02/scatter_probe_v10.txt), output
// identical to rsx_scatter2_kernel's; rsx_scatter2_kernel 0.506-0.507 ms, the same structure with two atomics per key
// (rsx_scatter9_handoff.hpp, HANDOFF = false) 0.495-0.501 ms in the same runs:
//   16 waves, 32 Ki-key tiles, chain first:  0.494-0.499 ms; without global stores 0.309-0.315 (two atomics: 0.337)
//   16 waves, the digit threads stage first: 0.492-0.493 ms; wave 0 has staged after 2.4 k cycles, the chain is done at 7.7 k
//   8 waves, 16 Ki-key tiles, two workgroups per CU: 0.602 ms (chain done at 15.6-17.4 k cycles, 33-37 tiles deep; LB 16: 0.628)
//   NEXT_HIST (the next column's histogram counted in the pass, README.md:774-779): 0.505 against 0.499 ms, histogram correct
// The staging itself is four times cheaper than with a second atomic, and the pass is exactly as long as before: the phase
// now ends when the chain is resolved (5.9 k cycles), and what bounds the kernel with its stores is the CU's memory
// pipeline (a tile's 128 KiB of loads and 128 KiB of stores one after the other) plus the phases in which it idles.
//
// rsx_scatter10 -- rsx_scatter2_kernel's structure (one-shot workgroup, ticket order, 32 Ki-key tiles, keys only, whole tiles)
// with ONE returning LDS atomic per key: the count phase's atomic returns the key's rank inside its (wave, digit) run (kept in
// 16 bits, two per register); after the layout the staging position is start[wave][digit] + rank -- an LDS read instead of a
// second atomic.  CHAIN_FIRST = true: the digit threads resolve the look-back before they stage their own keys (as
// rsx_scatter2_kernel); false: they stage first (their look-back loads are in flight meanwhile) and resolve the chain then.
#pragma once

#include "rsx_scatter2.hpp"

namespace rsx {

template <typename KT, int LB_ = 8, int NWAVES_ = 16> struct Sc10Cfg {
	static constexpr int NWAVES = NWAVES_;
	static constexpr int BLOCK = NWAVES * 64;
	static constexpr int ELEM = sizeof(KT);
	static constexpr int KPT = 128 / ELEM;
	static constexpr int TILE = BLOCK * KPT;
	static constexpr int LB = LB_;
	static constexpr int SB = 8;
	static constexpr int CHUNK = 16 / ELEM;
	static constexpr int STAGE_BYTES = TILE * ELEM;
};

template <typename KT, typename ST, typename C> struct Sc10Smem {
	__attribute__((aligned(16))) unsigned char stage_raw[C::STAGE_BYTES];
	u32 cell[C::NWAVES][256];
	ST delta[256];
	u32 wsum[4];
	u32 ticket;
	u32 nh[256];   // NEXT_HIST: the tile's counts of the NEXT column's digits (README.md:774-779, the histogram fused into the pass)
};

template <typename KT, typename ST, typename C = Sc10Cfg<KT>, bool TL = false, int DIG = DIG_GENERIC, bool CHAIN_FIRST = true, bool NEXT_HIST = false>
__global__ __launch_bounds__(C::BLOCK) void rsx_scatter10_kernel(const KT *__restrict__ kin, KT *__restrict__ kout, u32 ntiles, u32 shift,
                                                                 const u64 *__restrict__ gbase, ST *status, u32 *ticket, KdfArgs<KT> ka,
                                                                 u32 flags, u64 *tl, unsigned long long *next_hist = nullptr)
{
	typedef StatusBits<ST> SB_;
	constexpr int NWAVES = C::NWAVES, BLOCK = C::BLOCK, KPT = C::KPT, SB = C::SB, CHUNK = C::CHUNK, LB = C::LB;
	__shared__ Sc10Smem<KT, ST, C> sm;
	const u32 tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
	const u64 t_start = TL ? __builtin_readcyclecounter() : 0;
	if (tid == 0)
		sm.ticket = atomicAdd(ticket, 1u);
	for (u32 i = tid; i < NWAVES * 256; i += BLOCK)
		(&sm.cell[0][0])[i] = 0;
	if (NEXT_HIST && tid < 256)
		sm.nh[tid] = 0;
	__syncthreads();
	const u32 tile = __builtin_amdgcn_readfirstlane(sm.ticket);
	if (tile >= ntiles)
		return;
	auto opaque = [](u32 x) {
		asm volatile("" : "+v"(x));
		return x;
	};
	u32 *wc = sm.cell[wid];
	KT *stage_k = (KT *)sm.stage_raw;

	// ---- load (element loads: memory order) + rank inside the (wave, digit) run
	KT keep[KPT];
	u32 rk[KPT / 2];
	{
		const KT *p = kin + (u64)tile * C::TILE + (wid * (64 * KPT) + lane);
#pragma unroll
		for (int r = 0; r < KPT; ++r)
			keep[r] = p[r * 64];
#pragma unroll
		for (int r = 0; r < KPT; r += 2) {
			const u32 a = __hip_atomic_fetch_add(&wc[digit2<DIG>(keep[r], ka, shift)], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
			const u32 b = __hip_atomic_fetch_add(&wc[digit2<DIG>(keep[r + 1], ka, shift)], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
			rk[r / 2] = a | (b << 16);
		}
		if constexpr (NEXT_HIST) {
#pragma unroll
			for (int r = 0; r < KPT; ++r)
				atomicAdd(&sm.nh[digit2<DIG>(keep[r], ka, shift + 8)], 1u);
		}
	}
	__syncthreads();
	if (TL && tid == 0)
		tl[(u64)tile * 16 + 1] = __builtin_readcyclecounter();

	u32 tc = 0, incl = 0, tb = 0;
	ST w[LB];
	int back = (int)tile - 1;
	ST *my_status = status + (tile * 256u + tid);
	auto look = [&]() {
		const u32 t = opaque(tid);
#pragma unroll
		for (int j = 0; j < LB; ++j) {
			const int p = back - j > 0 ? back - j : 0;
			w[j] = __hip_atomic_load(status + ((u32)p * 256u + t), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		}
	};
	if (tid < 256) {
#pragma unroll
		for (int k = 0; k < NWAVES; ++k)
			tc += sm.cell[k][tid];
		const ST word = ((ST)(tile == 0 ? ST_PREFIX : ST_AGGREGATE) << SB_::SHIFT) | (ST)tc;
		__hip_atomic_store(my_status, word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		if (tile != 0)
			look();
		u32 x = tc;
#pragma unroll
		for (int off = 1; off < 64; off <<= 1) {
			const u32 y = __shfl_up(x, off);
			if (lane >= (u32)off)
				x += y;
		}
		incl = x;
		if (lane == 63)
			sm.wsum[opaque(wid)] = x;
	}
	__syncthreads();
	if (tid < 256) {
		tb = incl - tc;
		for (u32 k = 0; k < wid; ++k)
			tb += sm.wsum[k];
		u32 acc = tb;
#pragma unroll
		for (int k = 0; k < NWAVES; ++k) {
			const u32 c = sm.cell[k][tid];
			sm.cell[k][tid] = acc;
			acc += c;
		}
	}
	__syncthreads();
	if (TL && tid == 0) {
		tl[(u64)tile * 16 + 0] = t_start;
		tl[(u64)tile * 16 + 2] = __builtin_readcyclecounter();
	}
	auto chain = [&]() {
		u64 excl = 0;
		u32 depth = 0;
		if (tile != 0) {
			for (;;) {
				bool done = false;
				int used = 0;
#pragma unroll
				for (int j = 0; j < LB; ++j) {
					const u32 f = (u32)(w[j] >> SB_::SHIFT);
					if (!done && used == j && f != ST_EMPTY) {
						excl += (u64)(w[j] & SB_::VALMASK);
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
				look();
			}
			__hip_atomic_store(my_status, ((ST)ST_PREFIX << SB_::SHIFT) | (ST)(excl + tc), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		}
		sm.delta[tid] = (ST)(gbase[tid] + excl - tb);
		if (TL && tid == 0) {
			tl[(u64)tile * 16 + 3] = __builtin_readcyclecounter();
			tl[(u64)tile * 16 + 12] = depth;
		}
	};
	auto stage = [&]() {
#pragma unroll
		for (int r0 = 0; r0 < KPT; r0 += SB) {
			u32 pos[SB];
#pragma unroll
			for (int r = 0; r < SB; ++r) {
				const u32 q = (rk[(r0 + r) / 2] >> (16 * ((r0 + r) & 1))) & 0xFFFFu;
				pos[r] = wc[digit2<DIG>(keep[r0 + r], ka, shift)] + q;
			}
#pragma unroll
			for (int r = 0; r < SB; ++r)
				stage_k[pos[r]] = keep[r0 + r];
		}
	};
	if (CHAIN_FIRST) {
		if (tid < 256)
			chain();
		stage();
	} else {
		stage();
		if (TL && tid == 0)
			tl[(u64)tile * 16 + 6] = __builtin_readcyclecounter();
		if (tid < 256)
			chain();
	}
	__syncthreads();
	if (TL && tid == 0)
		tl[(u64)tile * 16 + 4] = __builtin_readcyclecounter();

	const ST *delta = sm.delta;
	const bool nostore = TL && (flags & SCATTER_DBG_NOSTORE);
#pragma unroll
	for (int j = 0; j < KPT / CHUNK; ++j) {
		if (j % 4 == 0)
			__builtin_amdgcn_sched_barrier(0);
		const u32 i0 = opaque(CHUNK * tid) + CHUNK * j * BLOCK;
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
			d[e] = digit2<DIG>(kv[e], ka, shift);
		if (!nostore) {
			if (d[0] == d[CHUNK - 1]) {
				store_chunk<KT, CHUNK>(kout + (ST)(delta[d[0]] + i0), kv);
			} else {
#pragma unroll
				for (int e = 0; e < CHUNK; ++e)
					kout[(ST)(delta[d[e]] + i0 + e)] = kv[e];
			}
		}
	}
	if (NEXT_HIST && tid < 256 && sm.nh[tid] != 0)
		atomicAdd(next_hist + tid, (unsigned long long)sm.nh[tid]);
	if (TL && tid == 0)
		tl[(u64)tile * 16 + 5] = __builtin_readcyclecounter();
}

}  // namespace rsx
