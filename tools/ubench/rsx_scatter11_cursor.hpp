// EXPERIMENT (round 4), not part of the library: rsx_scatter8_pipelined.hpp (round 2's software-pipelined persistent pass) with the
// look-back chain REPLACED by one returning global atomic per digit and tile on a cursor word.
// Round 2 found the pipelined form lose to the one-shot kernel because of the chain: its retries and PREFIX stores queue behind
// 128 KiB of key loads, every tile publishes late and the look-back walks 22 tiles.  The MSB passes of a keys-only sort without
// a histogram do not need the chain's ORDER: their buckets go to leaves that sort them anyway, so the tiles of a
// digit may land in any order -- a tile reserves its place with atomicAdd(cursor[digit], count).  No status words, no retries,
// nothing that waits for another tile.  Output: partitioned by digit, unstable across tiles.
#pragma once

#include "rsx_scatter2.hpp"

namespace rsx {

template <typename KT, int LB_ = 24> struct Sc11Cfg {
	static constexpr int NWAVES = 16;
	static constexpr int BLOCK = NWAVES * 64;
	static constexpr int ELEM = sizeof(KT);
	static constexpr int KPT = 128 / ELEM;
	static constexpr int TILE = BLOCK * KPT;
	static constexpr int LB = LB_;
	static constexpr int SB = 8;
	static constexpr int CHUNK = 16 / ELEM;
	static constexpr int STAGE_BYTES = TILE * ELEM;
	static_assert(KPT % 2 == 0 && 64 * KPT <= 65536, "two 16-bit ranks per register");
};

template <typename KT, typename ST, typename C> struct Sc11Smem {
	__attribute__((aligned(16))) unsigned char stage_raw[C::STAGE_BYTES];
	u32 cell[C::NWAVES][256];           // per (wave, digit): count while ranking, then the run's tile-local start
	ST delta[256];
	u32 wsum[4];
	u32 ticket[2];
};

enum : u32 { SC11_STAGGER_SHIFT = 16 };   // flags bits 16-19: start stagger, units of ~640 cycles x (workgroup % 32)

template <typename KT, typename ST, typename C = Sc11Cfg<KT>, bool TL = false, int DIG = DIG_GENERIC>
__global__ __launch_bounds__(C::BLOCK) void rsx_scatter11_kernel(const KT *__restrict__ kin, KT *__restrict__ kout, u32 ntiles, u32 shift,
                                                                const u64 *__restrict__ gbase, u32 *__restrict__ cursor, u32 *ticket,
                                                                KdfArgs<KT> ka, u32 flags, u64 *tl)
{
	constexpr int NWAVES = C::NWAVES, BLOCK = C::BLOCK, KPT = C::KPT, SB = C::SB, CHUNK = C::CHUNK;
	__shared__ Sc11Smem<KT, ST, C> sm;
	const u32 tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
	auto opaque = [](u32 x) {
		asm volatile("" : "+v"(x));
		return x;
	};
	u32 *wc = sm.cell[wid];
	KT *stage_k = (KT *)sm.stage_raw;
	const u32 wofs = wid * (64 * KPT) + lane;
	constexpr u32 TICKET_TID = BLOCK - 64;

	{
		const u32 stag = ((flags >> SC11_STAGGER_SHIFT) & 15u) * (blockIdx.x & 31u);
		for (u32 i = 0; i < stag; ++i)
			__builtin_amdgcn_s_sleep(10);
	}
	// Tiles are handed out in start order (=> the look-back cannot deadlock): the first one here, every further one at the
	// top of the iteration before it is requested -- taking two at once would put a workgroup's second tile between the
	// first tiles of its neighbours.
	if (tid == 0)
		sm.ticket[0] = atomicAdd(ticket, 1u);
	__syncthreads();
	u32 cur = __builtin_amdgcn_readfirstlane(sm.ticket[0]);
	if (cur >= ntiles)
		return;
	KT keep[KPT], ahead[KPT];
	u32 rk[KPT / 2];
	// (the order of a wave's keys does not matter to an unstable pass: 16-byte loads, no transposition)
	auto load_part = [&](KT (&dst)[KPT], const u32 tile, const int v0, const int nv) {
		constexpr int VEC = 16 / sizeof(KT);
		typedef KT vec_t __attribute__((ext_vector_type(VEC)));
		const vec_t *p = (const vec_t *)(kin + (u64)tile * C::TILE + (u64)wid * (64 * KPT)) + lane;
#pragma unroll
		for (int i = v0; i < v0 + nv; ++i) {
			const vec_t x = p[i * 64];
#pragma unroll
			for (int e = 0; e < VEC; ++e)
				dst[i * VEC + e] = x[e];
		}
	};
	auto load_tile = [&](KT (&dst)[KPT], const u32 tile) { load_part(dst, tile, 0, KPT * (int)sizeof(KT) / 16); };
	auto zero_row = [&]() {
#pragma unroll
		for (int k = 0; k < 4; ++k)
			wc[lane + 64 * k] = 0;   // a wave's own row: its DS operations execute in order
	};
	// ranks of KPT keys, in memory order (round r, then lane): 16 bits each
	auto rank_some = [&](const KT (&src)[KPT], const int r0, const int cnt) {
#pragma unroll
		for (int r = r0; r < r0 + cnt; r += 2) {
			const u32 a = __hip_atomic_fetch_add(&wc[digit2<DIG>(src[r], ka, shift)], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
			const u32 b = __hip_atomic_fetch_add(&wc[digit2<DIG>(src[r + 1], ka, shift)], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
			rk[r / 2] = a | (b << 16);
		}
	};
	// Layout of tile `tile` (the cells hold the counts of its (wave, digit) runs) AND its chain, both before any key load of
	// the next tile is in the CU's memory queue: totals, aggregate published, LB predecessors' status words requested; the
	// cells turned into tile-local run starts while those are on their way; then the look-back is resolved (further steps,
	// if any, find an empty queue too), the inclusive prefix published and the digits' global offsets left in sm.delta.
	auto layout = [&](const u32 tile, const u32 tk) {
		u32 incl = 0, tc = 0, tb = 0, base = 0;
		if (tid < 256) {
#pragma unroll
			for (int k = 0; k < NWAVES; ++k)
				tc += sm.cell[k][tid];
			// the tile's place among the digit's keys: whoever comes first (no order between tiles)
			base = __hip_atomic_fetch_add(cursor + tid, tc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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
			sm.delta[tid] = (ST)(gbase[tid] + base - tb);
			if (TL && tid == 0) {
				tl[(u64)tile * 16 + 12] = 0;
				tl[(u64)tile * 16 + 13] = 0;
			}
		}
		if (tid == TICKET_TID)
			sm.ticket[1] = tk;   // (requested at the top of the iteration: back by now)
		__syncthreads();
	};

	// ---- prologue: the first tile loaded and ranked
	load_tile(keep, cur);
	zero_row();
	rank_some(keep, 0, KPT);
	__syncthreads();

	for (u32 it = 0;; ++it) {
		const u64 t_start = TL ? __builtin_readcyclecounter() : 0;
		u32 tk = 0;
		if (tid == TICKET_TID) {
			typedef __attribute__((address_space(1))) u32 global_u32;
			global_u32 *tp = (global_u32 *)ticket;
			asm volatile("" : "+v"(tp));
			tk = __hip_atomic_fetch_add(tp, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		}
		// ---- 1. layout + chain of cur
		layout(cur, tk);
		const u32 nxt = __builtin_amdgcn_readfirstlane(sm.ticket[1]);
		const bool more = nxt < ntiles;
		if (TL && tid == 0) {
			tl[(u64)cur * 16 + 0] = t_start;
			tl[(u64)cur * 16 + 1] = __builtin_readcyclecounter();
		}
		// ---- 2. the next tile's keys requested; cur staged meanwhile: position = run start + rank
		if (more)
			load_tile(ahead, nxt);
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
		if (TL && tid == 0)
			tl[(u64)cur * 16 + 2] = __builtin_readcyclecounter();
		__syncthreads();   // staged
		if (TL && tid == 0)
			tl[(u64)cur * 16 + 3] = __builtin_readcyclecounter();

		// ---- 3. write-out of cur, interleaved with the ranking of next
		if (more)
			zero_row();
		const ST *delta = sm.delta;
		constexpr int NCH = KPT / CHUNK;       // chunks per lane
		constexpr int RPC = KPT / NCH;         // keys of next ranked per chunk written
#pragma unroll
		for (int j = 0; j < NCH; ++j) {
			if (more)
				rank_some(ahead, j * RPC, RPC);
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
				if (d[0] == d[CHUNK - 1]) {
					store_chunk<KT, CHUNK>(kout + (ST)(delta[d[0]] + i0), kv);
				} else {
#pragma unroll
					for (int e = 0; e < CHUNK; ++e)
						kout[(ST)(delta[d[e]] + i0 + e)] = kv[e];
				}
			}
		}
		__syncthreads();   // staging area read; next ranked
		if (TL && tid == 0)
			tl[(u64)cur * 16 + 4] = __builtin_readcyclecounter();
		if (!more)
			break;
#pragma unroll
		for (int r = 0; r < KPT; ++r)
			keep[r] = ahead[r];
		cur = nxt;
	}
}

}  // namespace rsx
