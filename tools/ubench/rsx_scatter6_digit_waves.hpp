// EXPERIMENT (round 2), not part of the library.  Measured on 2^28 u32 keys (profiles/r02/scatter_probe_all_experiments.txt):
// 0.528 ms per pass against 0.506 for rsx_scatter2_kernel.  The chain (6.0 k cycles) is hidden as intended, but the twelve
// key waves need 8.0 k cycles to rank and stage 40 keys per lane: the staging phase is bound by the LDS (one returning
// atomic and one store per key at random addresses: about 17 cycles per 64 keys with their bank conflicts), not by the chain.
// rsx_scatter6.hpp -- the scatter pass (radix_sort.hpp:82-90) with DEDICATED digit waves: keys only, gfx950.
//
// In rsx_scatter2_kernel the 256 digit threads (waves 0-3) resolve the look-back chain and THEN rank and stage their own
// keys, while waves 4-15 stage theirs at once: the staging phase lasts chain + a wave's staging (5.7 k + 3.1 k cycles of
// a tile's 30 k on 2^28 u32 keys).  Here waves 0-3 hold no keys at all: the tile is 12 waves x 64 lanes x 40 keys
// (30 Ki keys, 120 KiB of staging; the run of a digit is 480 bytes instead of 512), the digit waves publish, look back,
// lay out and resolve the chain beside the twelve key waves' staging, and the phase lasts max(chain, staging).  Keys are
// read with element loads (lane l of round r loads element 64 r + l: memory order without the transposition through the
// LDS); all sixteen waves write out.
//
// Tickets, status words, look-back, ranking by returning LDS atomics in memory order (see rsx_scatter2.hpp for what that
// rests on), partial last tile, device-side plan: as rsx_scatter2_kernel.
#pragma once

#include "rsx_scatter2.hpp"

namespace rsx {

template <typename KT, int LB_ = 8, int KPT_ = 0> struct Sc6Cfg {
	static constexpr int NWAVES = 16;                  // waves of the workgroup
	static constexpr int DW = 4;                       // digit waves (no keys)
	static constexpr int KW = NWAVES - DW;             // key waves
	static constexpr int BLOCK = NWAVES * 64;
	static constexpr int ELEM = sizeof(KT);
	static constexpr int KPT = KPT_ ? KPT_ : 160 / ELEM;   // keys per lane of a key wave: 120 KiB of staging
	static constexpr int TILE = KW * 64 * KPT;
	static constexpr int LB = LB_;
	static constexpr int SB = KPT % 8 == 0 ? 8 : (KPT % 5 == 0 ? 5 : 4);   // keys per lane ranked per batch
	static constexpr int CHUNK = 16 / ELEM;            // consecutive staged elements one lane writes out together
	static constexpr int NCHUNK = (TILE / CHUNK + BLOCK - 1) / BLOCK;   // write-out chunks per lane (the last round may be short)
	static constexpr int STAGE_BYTES = TILE * ELEM;
	static_assert(ELEM >= 4, "4- and 8-byte keys");
	static_assert(KPT % SB == 0 && TILE % CHUNK == 0, "whole batches / chunks");
};

template <typename KT, typename ST, typename C> struct Sc6Smem {
	__attribute__((aligned(16))) unsigned char stage_raw[C::STAGE_BYTES];
	u32 cell[C::KW][256];               // per (key wave, digit): count, then run start / cursor
	ST delta[256];                      // global offset of a digit's run minus its tile-local offset
	u32 wsum[4];
	u32 ticket;
};

template <typename KT, typename ST, typename C = Sc6Cfg<KT>, bool TL = false, int DIG = DIG_GENERIC>
__global__ __launch_bounds__(C::BLOCK) void rsx_scatter6_kernel(const KT *__restrict__ kin, KT *__restrict__ kout, u64 n, u32 shift,
                                                                const u64 *__restrict__ gbase, ST *status, u32 *ticket,
                                                                KdfArgs<KT> ka, u32 flags, u64 *tl,
                                                                const Plan *__restrict__ dplan = nullptr, u32 pass_index = 0)
{
	typedef StatusBits<ST> SB_;
	constexpr int KW = C::KW, DW = C::DW, BLOCK = C::BLOCK, KPT = C::KPT, SB = C::SB, CHUNK = C::CHUNK, LB = C::LB;
	// Device-scheduled pass (see rsx_scatter2_kernel): column, buffers and "nothing to do" from the device-side plan.
	if (dplan) {
		if (dplan->sorted || pass_index >= dplan->ncols)
			return;
		const u32 col = dplan->cols[pass_index];
		shift = 8 * col;
		gbase += 256 * col;
		if (pass_index & 1) {
			const KT *t = kin;
			kin = kout;
			kout = const_cast<KT *>(t);
		}
	}
	__shared__ Sc6Smem<KT, ST, C> sm;
	const u32 tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
	const u64 t_start = TL ? __builtin_readcyclecounter() : 0;
	if (tid == 0)
		sm.ticket = atomicAdd(ticket, 1u);   // tiles are handed out in start order => look-back cannot deadlock
	for (u32 i = tid; i < KW * 256; i += BLOCK)
		(&sm.cell[0][0])[i] = 0;
	__syncthreads();
	const u32 tile = __builtin_amdgcn_readfirstlane(sm.ticket);
	const u64 base = (u64)tile * C::TILE;
	if (base >= n)
		return;
	const u32 cnt = (n - base) < (u64)C::TILE ? (u32)(n - base) : (u32)C::TILE;
	const bool full = cnt == (u32)C::TILE;
	const bool keyw = wid >= (u32)DW;            // a key wave (wave-uniform)
	const u32 kwid = keyw ? wid - DW : 0u;
	const u32 wofs = kwid * (64 * KPT) + lane;   // key wave w owns [w*64*KPT, +64*KPT) of the tile; round r: element 64 r + lane
	auto opaque = [](u32 x) {
		asm volatile("" : "+v"(x));
		return x;
	};
	u32 *wc = sm.cell[kwid];
	KT *stage_k = (KT *)sm.stage_raw;

	// ---- key waves: load (element loads: memory order) + count; the keys stay in registers
	KT keep[KPT];
	if (keyw) {
		if (full) {
			const KT *p = kin + base + wofs;
#pragma unroll
			for (int r = 0; r < KPT; ++r)
				keep[r] = p[r * 64];
#pragma unroll
			for (int r = 0; r < KPT; ++r)
				atomicAdd(&wc[digit2<DIG>(keep[r], ka, shift)], 1u);
		} else {
			const u32 wo = opaque(wofs);
			const KT *p = kin + base;
#pragma unroll
			for (int r = 0; r < KPT; ++r) {
				const u32 o = wo + r * 64;
				keep[r] = o < cnt ? p[o] : (KT)0;
			}
#pragma unroll
			for (int r = 0; r < KPT; ++r) {
				const u32 o = wo + r * 64;
				if (o < cnt)
					atomicAdd(&wc[digit2<DIG>(keep[r], ka, shift)], 1u);
			}
		}
	}
	__syncthreads();   // #1
	if (TL && tid == 0)
		tl[(u64)tile * 16 + 1] = __builtin_readcyclecounter();

	// ---- digit waves (thread = digit): totals, publish the aggregate, START the look-back, layout
	u32 tc = 0, incl = 0, tb = 0;
	ST w[LB];
	int back = (int)tile - 1;   // nearest predecessor not consumed yet
	ST *my_status = status + (tile * 256u + tid);
	auto look = [&]() {
		const u32 t = opaque(tid);
#pragma unroll
		for (int j = 0; j < LB; ++j) {
			const int p = back - j > 0 ? back - j : 0;   // tile 0 always holds a prefix: safe filler
			w[j] = __hip_atomic_load(status + ((u32)p * 256u + t), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		}
	};
	if (!keyw) {
#pragma unroll
		for (int k = 0; k < KW; ++k)
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
	__syncthreads();   // #2
	if (!keyw) {
		tb = incl - tc;
		for (u32 k = 0; k < wid; ++k)
			tb += sm.wsum[k];
		u32 acc = tb;   // counts -> run starts, in place
#pragma unroll
		for (int k = 0; k < KW; ++k) {
			const u32 c = sm.cell[k][tid];
			sm.cell[k][tid] = acc;
			acc += c;
		}
	}
	__syncthreads();   // #3
	if (TL && tid == 0) {
		tl[(u64)tile * 16 + 0] = t_start;
		tl[(u64)tile * 16 + 2] = __builtin_readcyclecounter();
	}

	if (!keyw) {
		// ---- the chain: aggregates are summed until the first inclusive prefix; an empty word ends the batch
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
			const ST pword = ((ST)ST_PREFIX << SB_::SHIFT) | (ST)(excl + tc);
			__hip_atomic_store(my_status, pword, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		}
		sm.delta[tid] = (ST)(gbase[tid] + excl - tb);   // modulo 2^32 when ST is 32-bit (n < 2^30 then)
		if (TL && tid == 0) {
			tl[(u64)tile * 16 + 3] = __builtin_readcyclecounter();
			tl[(u64)tile * 16 + 12] = depth;
		}
	} else {
		// ---- rank + stage: the returning atomic on the (wave, digit) cursor is the key's tile-local position.  Rounds are
		// issued in memory order; lanes of a round come back in lane order.
#pragma unroll
		for (int r0 = 0; r0 < KPT; r0 += SB) {
			u32 pos[SB];
#pragma unroll
			for (int r = 0; r < SB; ++r) {
				pos[r] = 0;
				if (full || wofs + (r0 + r) * 64 < cnt)
					pos[r] = __hip_atomic_fetch_add(&wc[digit2<DIG>(keep[r0 + r], ka, shift)], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
			}
#pragma unroll
			for (int r = 0; r < SB; ++r)
				if (full || wofs + (r0 + r) * 64 < cnt)
					stage_k[pos[r]] = keep[r0 + r];
		}
		if (TL && tid == DW * 64)
			tl[(u64)tile * 16 + 6] = __builtin_readcyclecounter();   // a key wave through with its staging
	}
	__syncthreads();   // #4
	if (TL && tid == 0)
		tl[(u64)tile * 16 + 4] = __builtin_readcyclecounter();

	// ---- write-out (all waves): a lane takes CHUNK consecutive staged elements and, when they share a digit (first == last),
	// stores them with one wide store; chunks straddling a run boundary go element-wise
	const ST *delta = sm.delta;
#pragma unroll
	for (int j = 0; j < C::NCHUNK; ++j) {
		if (j % 4 == 0)
			__builtin_amdgcn_sched_barrier(0);   // keep a few chunks' registers alive at a time
		const u32 i0 = opaque(CHUNK * tid) + CHUNK * j * BLOCK;   // (recomputed: kept across the tile, the indices cost registers)
		if ((j + 1) * CHUNK * BLOCK > C::TILE && i0 >= (u32)C::TILE)
			break;   // (the short last round)
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
			const bool whole = full || i0 + CHUNK <= cnt;
			if (whole && d[0] == d[CHUNK - 1]) {
				store_chunk<KT, CHUNK>(kout + (ST)(delta[d[0]] + i0), kv);
			} else {
#pragma unroll
				for (int e = 0; e < CHUNK; ++e)
					if (full || i0 + e < cnt)
						kout[(ST)(delta[d[e]] + i0 + e)] = kv[e];
			}
		}
	}
	if (TL && tid == 0)
		tl[(u64)tile * 16 + 5] = __builtin_readcyclecounter();
}

}  // namespace rsx
