// EXPERIMENT (round 2), not part of the library; the carry described below was NOT built, because the first half of the
// idea already lost.  Measured on 2^28 u32 keys (profiles/r02/scatter_probe_all_experiments.txt): 0.570 ms per pass against 0.506 for
// rsx_scatter2_kernel.  Loading two tiles at once takes 35 k cycles, not 13 k: a CU pulls about 10 bytes per cycle from HBM
// however much it has in flight (MI355X_MICROARCH.md: ~10 B/cyc/CU), so the load phase is proportional to the bytes and the
// second tile's load is not hidden by the first's; the chain grows to 22 tiles (9.7 k cycles).
// rsx_scatter4.hpp -- the scatter pass (radix_sort.hpp:82-90) over PAIRS of tiles: keys only, whole tiles, gfx950.
//
// rsx_scatter2_kernel gives a workgroup one 32 Ki-key tile: load + count, publish, look back, rank + stage, write out --
// and the CU sees these phases strictly one after the other (0.506 ms per pass on 2^28 u32 keys, 0.345 ms of it without
// any global store: tools/ubench/scatter_probe.hip).  Two workgroups per CU with half the staging area each do not
// help (rsx_scatter3.hpp: the LDS is the shared bottleneck and the chain gets deeper).  What this kernel changes is what a
// workgroup amortises: it takes TWO consecutive tiles A and B,
//
//   * loads both at once (64 keys per lane in registers: twice the bytes in flight per CU while it waits for memory) and
//     counts both -- A into the 32-bit (wave, digit) cells, B into 16-bit ones --, publishes BOTH aggregates in the same
//     moment (status words stay per tile: successors and the table-ranked / single-tile kernels see the usual chain);
//   * looks back ONCE: B's exclusive prefix is A's plus A's counts, and B's inclusive prefix is published together with
//     A's, long before B is staged -- every second tile of the chain resolves without a round trip, which also shortens
//     the successors' walks;
//   * stages and writes A, then B through the same 128 KiB (run length unchanged: tile / 256), and -- CARRY -- hands the
//     ragged end of each of A's runs over to B inside the workgroup: the keys of A's run of digit d that fall into the
//     last, incomplete 64-byte atom of global memory are not written with A but kept (digit thread d, up to 15 registers)
//     and written together with the head of B's run of d as ONE whole atom.  HBM prices a partly written atom about
//     three times a whole one (tools/ubench/store_runs.hip), and every run of every tile has two of them; the carry
//     removes the two between A and B, a quarter of all.
//
// Stability, tickets, status-word format, look-back: as rsx_scatter2_kernel (see there for what ranking by returning LDS
// atomics rests on).  Only whole pairs are handled: the host gives the tail of the array (fewer than two tiles) to
// rsx_scatter2_kernel in the same chain (tile0 parameter).
#pragma once

#include "rsx_scatter2.hpp"

namespace rsx {

template <typename KT, int LB_ = 8, bool CARRY_ = true> struct Sc4Cfg {
	static constexpr int NWAVES = 16;
	static constexpr int BLOCK = NWAVES * 64;
	static constexpr int ELEM = sizeof(KT);
	static constexpr int KPT = 128 / ELEM;             // keys per lane and tile: 128 KiB of staging at 16 waves
	static constexpr int TILE = BLOCK * KPT;
	static constexpr int LB = LB_;
	static constexpr int SB = 8;                       // keys per lane ranked per batch
	static constexpr int CHUNK = 16 / ELEM;            // consecutive staged elements one lane writes out together
	static constexpr int ATOM = 64 / ELEM;             // keys per 64-byte atom of global memory
	static constexpr bool CARRY = CARRY_;
	static constexpr int STAGE_BYTES = TILE * ELEM;
	static_assert(TILE <= 32768, "B's counts and run starts live in 16-bit cells");
};

template <typename KT, typename ST, typename C> struct Sc4Smem {
	__attribute__((aligned(16))) unsigned char stage_raw[C::STAGE_BYTES];
	u32 cell[C::NWAVES][256];           // per (wave, digit): count, then run start / cursor (A, then B)
	u32 cellb[C::NWAVES][128];          // B's counts, 16 bits each, until A is done
	ST delta[256];                      // global offset of a digit's run minus its tile-local offset
	u32 wsum[4];
	u32 ticket;
};

template <typename KT, typename ST, typename C = Sc4Cfg<KT>, bool TL = false, int DIG = DIG_GENERIC>
__global__ __launch_bounds__(C::BLOCK) void rsx_scatter4_kernel(const KT *__restrict__ kin, KT *__restrict__ kout, u64 npairs, u32 shift,
                                                                const u64 *__restrict__ gbase, ST *status, u32 *ticket,
                                                                KdfArgs<KT> ka, u32 flags, u64 *tl,
                                                                const Plan *__restrict__ dplan = nullptr, u32 pass_index = 0)
{
	typedef StatusBits<ST> SB_;
	constexpr int NWAVES = C::NWAVES, BLOCK = C::BLOCK, KPT = C::KPT, SB = C::SB, CHUNK = C::CHUNK, LB = C::LB, ATOM = C::ATOM;
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
	__shared__ Sc4Smem<KT, ST, C> sm;
	const u32 tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
	const u64 t_start = TL ? __builtin_readcyclecounter() : 0;
	if (tid == 0)
		sm.ticket = atomicAdd(ticket, 1u);   // pairs are handed out in start order => look-back cannot deadlock
	for (u32 i = tid; i < NWAVES * 256; i += BLOCK)
		(&sm.cell[0][0])[i] = 0;
	for (u32 i = tid; i < NWAVES * 128; i += BLOCK)
		(&sm.cellb[0][0])[i] = 0;
	__syncthreads();
	const u32 pair = __builtin_amdgcn_readfirstlane(sm.ticket);
	if (pair >= npairs)
		return;
	const u32 tile_a = 2 * pair;
	const u64 base = (u64)tile_a * C::TILE;
	const u32 wofs = wid * (64 * KPT) + lane;   // wave w owns [w*64*KPT, +64*KPT) of a tile; round r: element 64 r + lane
	auto opaque = [](u32 x) {
		asm volatile("" : "+v"(x));
		return x;
	};
	u32 *wc = sm.cell[wid];
	u32 *wcb = sm.cellb[wid];
	KT *stage_k = (KT *)sm.stage_raw;

	// ---- load both tiles (element loads: a wave-instruction reads 64 consecutive keys, lane l of round r holds element
	// 64 r + l -- memory order, as the ranking needs it) and count: A into 32-bit cells, B into 16-bit ones
	KT ka_[KPT], kb_[KPT];
	{
		const KT *pa = kin + base + wofs;
		const KT *pb = pa + C::TILE;
#pragma unroll
		for (int r = 0; r < KPT; ++r)
			ka_[r] = pa[r * 64];
#pragma unroll
		for (int r = 0; r < KPT; ++r)
			kb_[r] = pb[r * 64];
#pragma unroll
		for (int r = 0; r < KPT; ++r)
			atomicAdd(&wc[digit2<DIG>(ka_[r], ka, shift)], 1u);
#pragma unroll
		for (int r = 0; r < KPT; ++r) {
			const u32 d = digit2<DIG>(kb_[r], ka, shift);
			atomicAdd(&wcb[d >> 1], 1u << ((d & 1u) * 16u));
		}
	}
	__syncthreads();   // #1
	if (TL && tid == 0)
		tl[(u64)pair * 16 + 1] = __builtin_readcyclecounter();

	// ---- digit thread d: totals of both tiles, publish both aggregates, START the look-back (for A), layout of A
	const unsigned short *cellb16 = (const unsigned short *)&sm.cellb[0][0];   // [NWAVES][256]
	u32 tca = 0, tcb = 0, incl = 0, tb = 0;
	ST w[LB];
	int back = (int)tile_a - 1;   // nearest predecessor not consumed yet
	ST *my_status = status + (tile_a * 256u + tid);   // A's word; B's is 256 words further
	auto look = [&]() {
		const u32 t = opaque(tid);
#pragma unroll
		for (int j = 0; j < LB; ++j) {
			const int p = back - j > 0 ? back - j : 0;   // tile 0 always holds a prefix: safe filler
			w[j] = __hip_atomic_load(status + ((u32)p * 256u + t), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		}
	};
	// digit-thread prefix over the 256 digits of one tile's totals `c`: tile-local start of the digit's run
	auto digit_scan_begin = [&](const u32 c) {
		u32 x = c;
#pragma unroll
		for (int off = 1; off < 64; off <<= 1) {
			const u32 y = __shfl_up(x, off);
			if (lane >= (u32)off)
				x += y;
		}
		incl = x;
		if (lane == 63)
			sm.wsum[opaque(wid)] = x;
	};
	if (tid < 256) {
#pragma unroll
		for (int k = 0; k < NWAVES; ++k) {
			tca += sm.cell[k][tid];
			tcb += cellb16[k * 256 + tid];
		}
		const ST word_a = ((ST)(tile_a == 0 ? ST_PREFIX : ST_AGGREGATE) << SB_::SHIFT) | (ST)tca;
		__hip_atomic_store(my_status, word_a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		const ST word_b = ((ST)(tile_a == 0 ? ST_PREFIX : ST_AGGREGATE) << SB_::SHIFT) | (ST)(tile_a == 0 ? tca + tcb : tcb);
		__hip_atomic_store(my_status + 256, word_b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		if (tile_a != 0)
			look();
		digit_scan_begin(tca);
	}
	__syncthreads();   // #2
	if (tid < 256) {
		tb = incl - tca;
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
		tl[(u64)pair * 16 + 0] = t_start;
		tl[(u64)pair * 16 + 2] = __builtin_readcyclecounter();
	}

	// ---- the chain, once for the pair (digit threads, before they stage their own keys: the other twelve waves stage
	// meanwhile).  Aggregates are summed until the first inclusive prefix; an empty word ends the batch.
	u64 excl = 0;   // keys of digit tid in all tiles before A
	if (tid < 256) {
		u32 depth = 0;
		if (tile_a != 0) {
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
			// both inclusive prefixes at once: B's successors need not wait for B to be staged
			__hip_atomic_store(my_status + 256, ((ST)ST_PREFIX << SB_::SHIFT) | (ST)(excl + tca + tcb), __ATOMIC_RELAXED,
			                   __HIP_MEMORY_SCOPE_AGENT);
			__hip_atomic_store(my_status, ((ST)ST_PREFIX << SB_::SHIFT) | (ST)(excl + tca), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		}
		sm.delta[tid] = (ST)(gbase[tid] + excl - tb);   // modulo 2^32 when ST is 32-bit (n < 2^30 then)
		if (TL && tid == 0) {
			tl[(u64)pair * 16 + 3] = __builtin_readcyclecounter();
			tl[(u64)pair * 16 + 12] = depth;
		}
	}

	// rank + stage one tile's keys: the returning atomic on the (wave, digit) cursor is the key's tile-local position
	auto stage_tile = [&](const KT (&keys)[KPT]) {
#pragma unroll
		for (int r0 = 0; r0 < KPT; r0 += SB) {
			u32 pos[SB];
#pragma unroll
			for (int r = 0; r < SB; ++r)
				pos[r] = __hip_atomic_fetch_add(&wc[digit2<DIG>(keys[r0 + r], ka, shift)], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#pragma unroll
			for (int r = 0; r < SB; ++r)
				stage_k[pos[r]] = keys[r0 + r];
		}
	};
	// The staged tile is sorted by digit and consecutive staged elements of one digit go to consecutive addresses: a lane
	// takes CHUNK consecutive elements and, when they share a digit (first == last), stores them with one wide store;
	// chunks straddling a run boundary go element-wise.  lo[d] / hi[d] (CARRY): the staged positions of digit d that this
	// pass over the tile writes -- the rest belongs to whole atoms assembled by the digit threads.
	const ST *delta = sm.delta;
	auto write_tile = [&]() {
#pragma unroll
		for (int j = 0; j < KPT / CHUNK; ++j) {
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
				if (d[0] == d[CHUNK - 1]) {
					store_chunk<KT, CHUNK>(kout + (ST)(delta[d[0]] + i0), kv);
				} else {
#pragma unroll
					for (int e = 0; e < CHUNK; ++e)
						kout[(ST)(delta[d[e]] + i0 + e)] = kv[e];
				}
			}
		}
	};

	// ---- tile A
	stage_tile(ka_);
	__syncthreads();   // #4
	if (TL && tid == 0)
		tl[(u64)pair * 16 + 4] = __builtin_readcyclecounter();
	write_tile();
	__syncthreads();   // #5: A's staging has been read
	if (TL && tid == 0)
		tl[(u64)pair * 16 + 5] = __builtin_readcyclecounter();

	// ---- tile B: layout from the 16-bit counts into the 32-bit cells, offsets = A's + A's counts
	if (tid < 256)
		digit_scan_begin(tcb);
	__syncthreads();   // #6
	if (tid < 256) {
		tb = incl - tcb;
		for (u32 k = 0; k < wid; ++k)
			tb += sm.wsum[k];
		u32 acc = tb;
#pragma unroll
		for (int k = 0; k < NWAVES; ++k) {
			sm.cell[k][tid] = acc;
			acc += cellb16[k * 256 + tid];
		}
		sm.delta[tid] = (ST)(gbase[tid] + excl + tca - tb);
	}
	__syncthreads();   // #7
	if (TL && tid == 0)
		tl[(u64)pair * 16 + 6] = __builtin_readcyclecounter();
	stage_tile(kb_);
	__syncthreads();   // #8
	if (TL && tid == 0)
		tl[(u64)pair * 16 + 7] = __builtin_readcyclecounter();
	write_tile();
	if (TL && tid == 0)
		tl[(u64)pair * 16 + 8] = __builtin_readcyclecounter();
}

}  // namespace rsx
