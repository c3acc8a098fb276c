// EXPERIMENT (round 2), not part of the library.  Measured on 2^28 u32 keys (profiles/r02/scatter_probe_all_experiments.txt):
//   prefetch issued by each wave after its staging: wait + count 5.0 k cycles (from 13 k) but chain 16.9 k (from 5.7 k) --
//     the status words of the look-back queue behind the key loads in the CU's memory pipeline -- 0.556 ms;
//   prefetch issued behind the barrier, before the write-out's stores: the STORES queue behind the loads, write-out
//     25.6 k cycles (from 6.3 k), 0.594 ms;
//   the persistent loop alone (no prefetch): 0.591 ms.
// against 0.506 ms for the one-shot rsx_scatter2_kernel: a CU's vector memory operations are served in order, so inside
// one workgroup loads, look-back and stores cannot overlap each other; only separate workgroups could (and the LDS holds one).
// rsx_scatter5.hpp -- the scatter pass (radix_sort.hpp:82-90) as a PERSISTENT workgroup that prefetches: keys only,
// whole tiles, gfx950.
//
// What bounds rsx_scatter2_kernel (tools/ubench/scatter_probe.hip, 2^28 u32 keys, 0.506 ms per pass): a CU pulls about
// 10 bytes per cycle from HBM however many loads it has in flight (a tile's 128 KiB take 13 k cycles; a pair of tiles
// loaded together, rsx_scatter4.hpp, takes 35 k), its LDS work takes another 11 k and the write-out 6 k -- and with one
// tile per CU these run strictly one after the other.  This kernel overlaps the first with the other two: one workgroup
// per CU loops over tickets, and the NEXT tile's keys are requested into a second set of registers as soon as a wave has
// staged its keys of the current tile -- by then the look-back chain is resolved (round 1's attempt, which requested them at
// the START of the staging phase, put 96 KiB of loads in front of the chain's status words in the CU's memory queue and
// doubled the chain's latency).  The loads cross the memory system during the write-out; the stores of the write-out are
// younger than the loads, so waiting for the keys never waits for more stores than have drained anyway.
//
// Everything else -- tickets in start order, per-tile status words and the decoupled look-back, ranking by returning LDS
// atomics in memory order, 32 Ki-key tiles staged in 128 KiB -- is rsx_scatter2_kernel's; the two kernels share a chain,
// so the host gives the last, partial tile to rsx_scatter2_kernel (tile0 parameter).
#pragma once

#include "rsx_scatter2.hpp"

namespace rsx {

template <typename KT, int LB_ = 8> struct Sc5Cfg {
	static constexpr int NWAVES = 16;
	static constexpr int BLOCK = NWAVES * 64;
	static constexpr int ELEM = sizeof(KT);
	static constexpr int KPT = 128 / ELEM;             // keys per lane and tile: 128 KiB of staging at 16 waves
	static constexpr int TILE = BLOCK * KPT;
	static constexpr int LB = LB_;
	static constexpr int SB = 8;                       // keys per lane ranked per batch
	static constexpr int CHUNK = 16 / ELEM;            // consecutive staged elements one lane writes out together
	static constexpr int STAGE_BYTES = TILE * ELEM;
};

template <typename KT, typename ST, typename C> struct Sc5Smem {
	__attribute__((aligned(16))) unsigned char stage_raw[C::STAGE_BYTES];
	u32 cell[C::NWAVES][256];           // per (wave, digit): count, then run start / cursor
	ST delta[256];                      // global offset of a digit's run minus its tile-local offset
	u32 wsum[4];
	u32 ticket[2];
};

template <typename KT, typename ST, typename C = Sc5Cfg<KT>, bool TL = false, int DIG = DIG_GENERIC>
__global__ __launch_bounds__(C::BLOCK) void rsx_scatter5_kernel(const KT *__restrict__ kin, KT *__restrict__ kout, u32 ntiles, u32 shift,
                                                                const u64 *__restrict__ gbase, ST *status, u32 *ticket,
                                                                KdfArgs<KT> ka, u32 flags, u64 *tl,
                                                                const Plan *__restrict__ dplan = nullptr, u32 pass_index = 0)
{
	typedef StatusBits<ST> SB_;
	constexpr int NWAVES = C::NWAVES, BLOCK = C::BLOCK, KPT = C::KPT, SB = C::SB, CHUNK = C::CHUNK, LB = C::LB;
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
	__shared__ Sc5Smem<KT, ST, C> sm;
	const u32 tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
	auto opaque = [](u32 x) {
		asm volatile("" : "+v"(x));
		return x;
	};
	u32 *wc = sm.cell[wid];
	KT *stage_k = (KT *)sm.stage_raw;
	const u32 wofs = wid * (64 * KPT) + lane;   // wave w owns [w*64*KPT, +64*KPT) of a tile; round r: element 64 r + lane
	constexpr u32 TICKET_TID = BLOCK - 64;      // (a lane of the last wave: not a digit thread)

	// tickets: the current tile and the next one (tiles are handed out in start order => look-back cannot deadlock)
	if (tid == 0) {
		sm.ticket[0] = atomicAdd(ticket, 1u);
		sm.ticket[1] = atomicAdd(ticket, 1u);
	}
	__syncthreads();
	u32 cur = __builtin_amdgcn_readfirstlane(sm.ticket[0]);
	u32 nxt = __builtin_amdgcn_readfirstlane(sm.ticket[1]);
	if (cur >= ntiles)
		return;
	// element loads: a wave-instruction reads 64 consecutive keys, lane l of round r holds element 64 r + l (memory order)
	KT keep[KPT], ahead[KPT];
	auto load_tile = [&](KT (&dst)[KPT], const u32 tile) {
		const KT *p = kin + (u64)tile * C::TILE + wofs;
#pragma unroll
		for (int r = 0; r < KPT; ++r)
			dst[r] = p[r * 64];
	};
	load_tile(keep, cur);
	__syncthreads();   // sm.ticket read by everybody before it is reused

	for (u32 it = 0;; ++it) {
		const u64 t_start = TL ? __builtin_readcyclecounter() : 0;
		const bool more = nxt < ntiles;
		u32 tk = 0;
		if (tid == TICKET_TID && more) {   // the ticket after the next: back long before it is handed over
			typedef __attribute__((address_space(1))) u32 global_u32;
			global_u32 *tp = (global_u32 *)ticket;
			asm volatile("" : "+v"(tp));
			tk = __hip_atomic_fetch_add(tp, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		}
#pragma unroll
		for (int k = 0; k < 4; ++k)
			wc[lane + 64 * k] = 0;           // (a wave's own row: DS operations of a wave execute in order)
		// ---- count
#pragma unroll
		for (int r = 0; r < KPT; ++r)
			atomicAdd(&wc[digit2<DIG>(keep[r], ka, shift)], 1u);
		__syncthreads();   // #1
		if (TL && tid == 0)
			tl[(u64)cur * 16 + 1] = __builtin_readcyclecounter();

		// ---- digit thread d: totals, publish the aggregate, START the look-back, layout
		u32 tc = 0, incl = 0, tb = 0;
		ST w[LB];
		int back = (int)cur - 1;   // nearest predecessor not consumed yet
		ST *my_status = status + (cur * 256u + tid);
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
			const ST word = ((ST)(cur == 0 ? ST_PREFIX : ST_AGGREGATE) << SB_::SHIFT) | (ST)tc;
			__hip_atomic_store(my_status, word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			if (cur != 0)
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
			tl[(u64)cur * 16 + 0] = t_start;
			tl[(u64)cur * 16 + 2] = __builtin_readcyclecounter();
		}
		// ---- the chain (digit threads, before they stage their own keys: the other twelve waves stage meanwhile)
		if (tid < 256) {
			u64 excl = 0;
			u32 depth = 0;
			if (cur != 0) {
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
			sm.delta[tid] = (ST)(gbase[tid] + excl - tb);   // modulo 2^32 when ST is 32-bit (n < 2^30 then)
			if (TL && tid == 0) {
				tl[(u64)cur * 16 + 3] = __builtin_readcyclecounter();
				tl[(u64)cur * 16 + 12] = depth;
			}
		}
		// ---- rank + stage: the returning atomic on the (wave, digit) cursor is the key's tile-local position
#pragma unroll
		for (int r0 = 0; r0 < KPT; r0 += SB) {
			u32 pos[SB];
#pragma unroll
			for (int r = 0; r < SB; ++r)
				pos[r] = __hip_atomic_fetch_add(&wc[digit2<DIG>(keep[r0 + r], ka, shift)], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#pragma unroll
			for (int r = 0; r < SB; ++r)
				stage_k[pos[r]] = keep[r0 + r];
		}
		if (tid == TICKET_TID && more)
			sm.ticket[it & 1] = tk;
		__syncthreads();   // #4
		if (TL && tid == 0)
			tl[(u64)cur * 16 + 4] = __builtin_readcyclecounter();
		const u32 nn = more ? __builtin_amdgcn_readfirstlane(sm.ticket[it & 1]) : ntiles;
		// ---- the next tile's keys: requested now, behind the barrier -- every digit thread has resolved its chain (requested
		// while the chain is still looking back, measured: the status words queue behind the key loads in the CU's memory
		// pipeline and the chain takes three times as long) -- and before the write-out's stores, which are younger
		if (more && !(flags & SCATTER_DBG_LINEAR))
			load_tile(ahead, nxt);

		// ---- write-out: a lane takes CHUNK consecutive staged elements and, when they share a digit (first == last),
		// stores them with one wide store; chunks straddling a run boundary go element-wise
		const ST *delta = sm.delta;
#pragma unroll
		for (int j = 0; j < KPT / CHUNK; ++j) {
			if (j % 4 == 0)
				__builtin_amdgcn_sched_barrier(0);   // keep a few chunks' registers alive at a time
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
		__syncthreads();   // #5: the staging area has been read
		if (TL && tid == 0)
			tl[(u64)cur * 16 + 5] = __builtin_readcyclecounter();
		if (!more)
			break;
		if (flags & SCATTER_DBG_LINEAR)       // probe: no prefetch at all (the persistent loop alone)
			load_tile(ahead, nxt);
#pragma unroll
		for (int r = 0; r < KPT; ++r)
			keep[r] = ahead[r];
		cur = nxt;
		nxt = nn;
	}
}

}  // namespace rsx
