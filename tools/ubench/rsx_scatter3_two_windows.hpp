// EXPERIMENT (round 2), not part of the library.  Measured on 2^28 u32 keys (profiles/r02/scatter_probe_all_experiments.txt):
// 64 keys per lane spill (0.84 ms); 48 keys per lane (24 Ki-key tiles): 0.55-0.57 ms against 0.506 for rsx_scatter2_kernel,
// 0.361 ms without global stores (0.34): two workgroups per CU do not overlap anything that matters -- a key is offered to
// the LDS once more (the second, predicated staging pass), eight waves hide the LDS latency worse than sixteen, and the
// chain gets deeper (22 tiles, three round trips behind the digit waves' own 9 k cycles of ranking).
// rsx_scatter3.hpp -- the scatter pass (radix_sort.hpp:82-90) with TWO workgroups per CU: keys only, gfx950.
//
// rsx_scatter2_kernel keeps a 32 Ki-key tile's keys in registers, ranks them, stages the whole tile in LDS (128 KiB) and
// writes 512-byte runs.  One such workgroup fills a CU's LDS, so a CU goes through a tile's phases one after the other:
// while the keys are on their way the LDS idles, while the LDS ranks nothing is in flight, and the stores of the
// write-out drain before the next workgroup can start.  Measured on 2^28 u32 keys: 0.506 ms per pass, 0.345 ms with the
// global stores compiled out -- the pass is bound by the ORDER of its phases, not by the memory system alone.
//
// This kernel keeps the tile (same run length: with 256 digits the run is tile / 256) but stages it in two WINDOWS of
// half a tile: every key's tile-local position is computed once (one returning LDS atomic, as before) and remembered
// (16 bits, two per register); the keys whose position lies in the first half are staged and written out, then the
// others.  Staging takes 64 KiB, a workgroup has 8 waves (64 keys per lane), and TWO workgroups share a CU: one tile's
// loads, chain and stores overlap the other's LDS work without any software pipelining.  The price: every key is
// offered to the LDS twice for the staging store (half of the lanes are active each time) and 96 registers hold keys and
// positions.
//
// Everything else is rsx_scatter2_kernel's: tickets, status words and the decoupled look-back (same format, so that the
// two kernels could share a chain), ranking by returning LDS atomics in memory order (see there for what that rests on),
// whole tiles without bounds tests, the device-side plan for speculative passes.
#pragma once

#include "rsx_scatter2.hpp"

namespace rsx {

template <typename KT, int NWAVES_ = 8, int LB_ = 8, int KPT_ = 0, bool VLOAD_ = true> struct Sc3Cfg {
	static constexpr int NWAVES = NWAVES_;
	static constexpr int BLOCK = NWAVES * 64;
	static constexpr int ELEM = sizeof(KT);
	static constexpr int KPT = KPT_ ? KPT_ : 256 / ELEM;           // keys per lane: 128 KiB of keys per tile at 8 waves
	static constexpr int TILE = BLOCK * KPT;
	static constexpr int NWIN = 2;
	static constexpr int WIN = TILE / NWIN;                          // tile-local positions per window
	static constexpr int LB = LB_;
	static constexpr int SB = 8;                                     // keys per lane ranked per batch
	static constexpr int VEC = 16 / ELEM;
	static constexpr int CHUNK = 16 / ELEM;                          // consecutive staged elements one lane writes out together
	static constexpr bool VLOAD = VLOAD_ && sizeof(KT) >= 4;         // 16-byte loads + transposition through LDS, or element loads
	static constexpr int STAGE_BYTES = WIN * ELEM;
	static_assert(NWAVES >= 4, "256 digit threads are needed");
	static_assert(TILE <= 65536, "tile-local positions are kept in 16 bits");
	static_assert(KPT % SB == 0 && KPT % VEC == 0 && WIN % (CHUNK * BLOCK) == 0, "whole batches / vectors / chunks");
};

template <typename KT, typename ST, typename C> struct Sc3Smem {
	__attribute__((aligned(16))) unsigned char stage_raw[C::STAGE_BYTES];
	u32 cell[C::NWAVES][256];           // per (wave, digit): count, then run start / cursor
	ST delta[256];                      // global offset of a digit's run minus its tile-local offset
	u32 wsum[4];
	u32 ticket;
};

template <typename KT, typename ST, typename C = Sc3Cfg<KT>, bool TL = false, int DIG = DIG_GENERIC>
__global__ __launch_bounds__(C::BLOCK, 2 * C::BLOCK / 256) void rsx_scatter3_kernel(const KT *__restrict__ kin, KT *__restrict__ kout, u64 n,
                                                                                     u32 shift, const u64 *__restrict__ gbase, ST *status,
                                                                                     u32 *ticket, KdfArgs<KT> ka, u32 flags, u64 *tl,
                                                                                     const Plan *__restrict__ dplan = nullptr,
                                                                                     u32 pass_index = 0)
{
	typedef StatusBits<ST> SB_;
	constexpr int NWAVES = C::NWAVES, BLOCK = C::BLOCK, KPT = C::KPT, SB = C::SB, CHUNK = C::CHUNK, LB = C::LB, VEC = C::VEC;
	constexpr u32 WIN = C::WIN;
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
	__shared__ Sc3Smem<KT, ST, C> sm;
	const u32 tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
	const u64 t_start = TL ? __builtin_readcyclecounter() : 0;
	if (tid == 0)
		sm.ticket = atomicAdd(ticket, 1u);   // tiles are handed out in start order => look-back cannot deadlock
	for (u32 i = tid; i < NWAVES * 256; i += BLOCK)
		(&sm.cell[0][0])[i] = 0;
	__syncthreads();
	const u32 tile = __builtin_amdgcn_readfirstlane(sm.ticket);
	const u64 base = (u64)tile * C::TILE;
	if (base >= n)
		return;
	const u32 cnt = (n - base) < (u64)C::TILE ? (u32)(n - base) : (u32)C::TILE;
	const bool full = cnt == (u32)C::TILE;
	const u32 wofs = wid * (64 * KPT) + lane;   // wave w owns [w*64*KPT, +64*KPT) of the tile; round r: element 64 r + lane
	auto opaque = [](u32 x) {
		asm volatile("" : "+v"(x));
		return x;
	};
	u32 *wc = sm.cell[wid];
	KT *stage_k = (KT *)sm.stage_raw;

	// ---- load + count: the keys stay in registers
	KT keep[KPT];
	if (full && C::VLOAD && (((uintptr_t)kin) & 15) == 0) {
		typedef KT vec_t __attribute__((ext_vector_type(VEC)));
		constexpr int NV = KPT / VEC;   // 16-byte loads per lane, all in flight
		const vec_t *vp = (const vec_t *)(kin + base + (u64)wid * (64 * KPT)) + lane;
		vec_t v[NV];
#pragma unroll
		for (int i = 0; i < NV; ++i)
			v[i] = vp[i * 64];
		// 16-byte loads give lane l the elements VEC (64 i + l) .. + VEC - 1; ranking wants round r = element 64 r + l: transpose
		// through one KiB of the (still unused) staging area per wave.  DS operations of one wave execute in issue order,
		// so the same KiB serves every vector and no barrier is needed.
		KT *scratch = stage_k + (u32)wid * (64 * VEC);
#pragma unroll
		for (int i = 0; i < NV; ++i) {
			*((vec_t *)scratch + lane) = v[i];
			RSX_COMPILER_FENCE();
#pragma unroll
			for (int e = 0; e < VEC; ++e)
				keep[i * VEC + e] = scratch[e * 64 + lane];
			RSX_COMPILER_FENCE();
#pragma unroll
			for (int e = 0; e < VEC; ++e)
				atomicAdd(&wc[digit2<DIG>(keep[i * VEC + e], ka, shift)], 1u);
		}
	} else if (full) {
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
	__syncthreads();   // #1
	if (TL && tid == 0)
		tl[(u64)tile * 16 + 1] = __builtin_readcyclecounter();

	// ---- digit thread d: totals, publish the aggregate, START the look-back, layout
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
	__syncthreads();   // #2
	if (tid < 256) {
		tb = incl - tc;
		for (u32 k = 0; k < wid; ++k)
			tb += sm.wsum[k];
		u32 acc = tb;   // counts -> run starts, in place
#pragma unroll
		for (int k = 0; k < NWAVES; ++k) {
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

	// ---- rank every key once (the returning atomic on the (wave, digit) cursor is its tile-local position, remembered in
	// 16 bits) and stage the keys of window 0.  Rounds are issued in memory order; lanes of a round come back in lane order.
	u32 posp[KPT / 2];
#pragma unroll
	for (int r0 = 0; r0 < KPT; r0 += SB) {
		u32 pos[SB];
#pragma unroll
		for (int r = 0; r < SB; ++r) {
			pos[r] = 0xFFFFu;
			if (full || wofs + (r0 + r) * 64 < cnt)
				pos[r] = __hip_atomic_fetch_add(&wc[digit2<DIG>(keep[r0 + r], ka, shift)], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
		}
#pragma unroll
		for (int r = 0; r < SB; ++r)
			if (pos[r] < WIN)
				stage_k[pos[r]] = keep[r0 + r];
#pragma unroll
		for (int r = 0; r < SB; r += 2)
			posp[(r0 + r) >> 1] = pos[r] | (pos[r + 1] << 16);
	}
	// ---- the chain (digit threads, after their own staging: the first window of status words has arrived meanwhile):
	// aggregates are summed until the first inclusive prefix; an empty word ends the batch
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
			const ST pword = ((ST)ST_PREFIX << SB_::SHIFT) | (ST)(excl + tc);
			__hip_atomic_store(my_status, pword, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		}
		sm.delta[tid] = (ST)(gbase[tid] + excl - tb);   // modulo 2^32 when ST is 32-bit (n < 2^30 then)
		if (TL && tid == 0) {
			tl[(u64)tile * 16 + 3] = __builtin_readcyclecounter();
			tl[(u64)tile * 16 + 12] = depth;
		}
	}
	__syncthreads();   // #4
	if (TL && tid == 0)
		tl[(u64)tile * 16 + 4] = __builtin_readcyclecounter();

	// ---- write-out of one window.  The staged window is sorted by digit and consecutive staged elements of one digit go
	// to consecutive addresses: a lane takes CHUNK consecutive elements and, when they share a digit (first == last),
	// stores them with one wide store; chunks straddling a run boundary go element-wise.
	const ST *delta = sm.delta;
	auto write_window = [&](const u32 wbase) {
#pragma unroll
		for (int j = 0; j < (int)(WIN / (CHUNK * BLOCK)); ++j) {
			if (j % 4 == 0)
				__builtin_amdgcn_sched_barrier(0);   // keep a few chunks' registers alive at a time
			const u32 i0 = opaque(CHUNK * tid) + CHUNK * j * BLOCK;   // (recomputed: kept across the tile, the indices cost registers)
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
				const u32 p0 = wbase + i0;   // tile-local position of the chunk's first element
				const bool whole = full || p0 + CHUNK <= cnt;
				if (whole && d[0] == d[CHUNK - 1]) {
					store_chunk<KT, CHUNK>(kout + (ST)(delta[d[0]] + p0), kv);
				} else {
#pragma unroll
					for (int e = 0; e < CHUNK; ++e)
						if (full || p0 + e < cnt)
							kout[(ST)(delta[d[e]] + p0 + e)] = kv[e];
				}
			}
		}
	};
	write_window(0);
	if (!full && cnt <= WIN)
		return;            // (uniform: a partial tile that fits the first window)
	__syncthreads();   // #5: window 0 has been read
	if (TL && tid == 0)
		tl[(u64)tile * 16 + 5] = __builtin_readcyclecounter();
	// ---- window 1: the keys whose position is WIN or more (0xFFFF: no key)
#pragma unroll
	for (int r = 0; r < KPT; ++r) {
		const u32 p = (posp[r >> 1] >> (16 * (r & 1))) & 0xFFFFu;
		if (p >= WIN && p != 0xFFFFu)
			stage_k[p - WIN] = keep[r];
	}
	__syncthreads();   // #6
	if (TL && tid == 0)
		tl[(u64)tile * 16 + 6] = __builtin_readcyclecounter();
	write_window(WIN);
	if (TL && tid == 0)
		tl[(u64)tile * 16 + 7] = __builtin_readcyclecounter();
}

}  // namespace rsx
