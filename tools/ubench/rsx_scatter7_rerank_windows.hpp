// EXPERIMENT (round 2), not part of the library.  Measured on 2^28 u32 keys (profiles/r02/scatter_probe_all_experiments.txt):
// 0.537-0.561 ms per pass against 0.49-0.51 for rsx_scatter2_kernel (0.374 ms without global stores against 0.34): load +
// count 20.6 k cycles, chain 14.7 k (21 tiles deep), and every key goes to the LDS five times instead of three.
//
// rsx_scatter7 -- two workgroups of 16 waves per CU, 32 Ki-key tiles (run length unchanged), staged in two WINDOWS of
// 64 KiB, the keys in 32 registers per lane and NO remembered positions: each window re-ranks all keys (a second
// returning LDS atomic per key; the cursors are put back to the run starts between the windows) and stages those whose
// position falls into it.  16-bit cells (two digits per word) so that a workgroup needs 73 KiB of LDS; at most 64 registers
// so that two workgroups of 16 waves share a CU.  The bet: with two tiles in flight per CU one tile's loads, chain and
// stores overlap the other's LDS work; the price is a third LDS atomic and a second (predicated) staging store per key.
#pragma once

#include "rsx_scatter2.hpp"

namespace rsx {

template <typename KT, int LB_ = 8> struct Sc7Cfg {
	static constexpr int NWAVES = 16;
	static constexpr int BLOCK = NWAVES * 64;
	static constexpr int ELEM = sizeof(KT);
	static constexpr int KPT = 128 / ELEM;             // keys per lane: 128 KiB of keys per tile
	static constexpr int TILE = BLOCK * KPT;
	static constexpr int NWIN = 2;
	static constexpr int WIN = TILE / NWIN;
	static constexpr int LB = LB_;
	static constexpr int SB = 8;
	static constexpr int CHUNK = 16 / ELEM;
	static constexpr int STAGE_BYTES = WIN * ELEM;
	static_assert(TILE <= 32768, "16-bit cells");
	static_assert(WIN % (CHUNK * BLOCK) == 0, "whole chunks per lane and window");
};

template <typename KT, typename ST, typename C> struct Sc7Smem {
	__attribute__((aligned(16))) unsigned char stage_raw[C::STAGE_BYTES];
	u32 cell[C::NWAVES][128];           // per (wave, digit), 16 bits each: count, then run start / cursor
	ST delta[256];
	u32 wsum[4];
	u32 ticket;
};

template <typename KT, typename ST, typename C = Sc7Cfg<KT>, bool TL = false, int DIG = DIG_GENERIC>
__global__ __launch_bounds__(C::BLOCK, 8) void rsx_scatter7_kernel(const KT *__restrict__ kin, KT *__restrict__ kout, u64 n, u32 shift,
                                                                   const u64 *__restrict__ gbase, ST *status, u32 *ticket,
                                                                   KdfArgs<KT> ka, u32 flags, u64 *tl)
{
	typedef StatusBits<ST> SB_;
	constexpr int NWAVES = C::NWAVES, BLOCK = C::BLOCK, KPT = C::KPT, SB = C::SB, CHUNK = C::CHUNK, LB = C::LB;
	constexpr u32 WIN = C::WIN;
	__shared__ Sc7Smem<KT, ST, C> sm;
	const u32 tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
	const u64 t_start = TL ? __builtin_readcyclecounter() : 0;
	if (tid == 0)
		sm.ticket = atomicAdd(ticket, 1u);
	for (u32 i = tid; i < NWAVES * 128; i += BLOCK)
		(&sm.cell[0][0])[i] = 0;
	__syncthreads();
	const u32 tile = __builtin_amdgcn_readfirstlane(sm.ticket);
	const u64 base = (u64)tile * C::TILE;
	if (base + C::TILE > n)
		return;   // (probe: whole tiles only)
	auto opaque = [](u32 x) {
		asm volatile("" : "+v"(x));
		return x;
	};
	u32 *wc = sm.cell[wid];
	KT *stage_k = (KT *)sm.stage_raw;
	unsigned short *cell16 = (unsigned short *)&sm.cell[0][0];   // [NWAVES][256]

	// ---- load (element loads: memory order) + count
	KT keep[KPT];
	{
		const KT *p = kin + base + (wid * (64 * KPT) + lane);
#pragma unroll
		for (int r = 0; r < KPT; ++r)
			keep[r] = p[r * 64];
#pragma unroll
		for (int r = 0; r < KPT; ++r) {
			const u32 d = digit2<DIG>(keep[r], ka, shift);
			atomicAdd(&wc[d >> 1], 1u << ((d & 1u) * 16u));
		}
	}
	__syncthreads();   // #1
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
			tc += cell16[k * 256 + tid];
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
	__syncthreads();   // #2
	if (tid < 256) {
		tb = incl - tc;
		for (u32 k = 0; k < wid; ++k)
			tb += sm.wsum[k];
		u32 acc = tb;   // counts -> run starts, in place
#pragma unroll
		for (int k = 0; k < NWAVES; ++k) {
			const u32 c = cell16[k * 256 + tid];
			cell16[k * 256 + tid] = (unsigned short)acc;
			acc += c;
		}
	}
	__syncthreads();   // #3
	if (TL && tid == 0) {
		tl[(u64)tile * 16 + 0] = t_start;
		tl[(u64)tile * 16 + 2] = __builtin_readcyclecounter();
	}
	// ---- the chain (digit threads first, as rsx_scatter2_kernel)
	if (tid < 256) {
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
	}
	const ST *delta = sm.delta;
#pragma unroll 1
	for (u32 win = 0; win < (u32)C::NWIN; ++win) {
		const u32 wbase = win * WIN;
		// (an opaque copy per window: otherwise the digits, word addresses and shifts of all 32 keys are hoisted out of the
		// window loop and spilled)
#pragma unroll
		for (int r = 0; r < KPT; ++r)
			asm volatile("" : "+v"(keep[r]));
		// ---- rank every key (returning atomic on the 16-bit cursor), stage those of this window
#pragma unroll
		for (int r0 = 0; r0 < KPT; r0 += SB) {
			u32 pos[SB];
#pragma unroll
			for (int r = 0; r < SB; ++r) {
				const u32 d = digit2<DIG>(keep[r0 + r], ka, shift);
				const u32 sh = (d & 1u) * 16u;
				const u32 old = __hip_atomic_fetch_add(&wc[d >> 1], 1u << sh, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
				pos[r] = ((old >> sh) & 0xFFFFu) - wbase;
			}
#pragma unroll
			for (int r = 0; r < SB; ++r)
				if (pos[r] < WIN)
					stage_k[pos[r]] = keep[r0 + r];
		}
		__syncthreads();   // #4
		if (TL && tid == 0)
			tl[(u64)tile * 16 + 4 + 2 * win] = __builtin_readcyclecounter();
		if (win + 1 < (u32)C::NWIN && tid < 256) {
			// the cursors back to the run starts: (wave k, digit)'s start is where (wave k - 1, digit) ended
			u32 prev = tb;
#pragma unroll
			for (int k = 0; k < NWAVES; ++k) {
				const u32 endk = cell16[k * 256 + tid];
				cell16[k * 256 + tid] = (unsigned short)prev;
				prev = endk;
			}
		}
		// ---- write-out of the window
#pragma unroll
		for (int j = 0; j < (int)(WIN / (CHUNK * BLOCK)); ++j) {
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
			if (!(TL && (flags & SCATTER_DBG_NOSTORE))) {
				const u32 p0 = wbase + i0;
				if (d[0] == d[CHUNK - 1]) {
					store_chunk<KT, CHUNK>(kout + (ST)(delta[d[0]] + p0), kv);
				} else {
#pragma unroll
					for (int e = 0; e < CHUNK; ++e)
						kout[(ST)(delta[d[e]] + p0 + e)] = kv[e];
				}
			}
		}
		__syncthreads();   // #5: the window has been read (and the cursors are reset)
		if (TL && tid == 0)
			tl[(u64)tile * 16 + 5 + 2 * win] = __builtin_readcyclecounter();
	}
}

}  // namespace rsx
