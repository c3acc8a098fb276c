// rsx_scatter5_experimental.hpp -- EXPERIMENT, not part of the library: rsx_scatter2_kernel's tile body in a persistent
// workgroup (one per CU) that loads the NEXT tile's keys at the start of the staging phase: the loads cross the CU's
// memory queue while the LDS does the ranking, are in registers before the write-out's stores enter the queue, and the
// stores drain while the next tile is counted.  The digit threads issue their loads after they have resolved the chain
// (per wave, vector-memory operations complete in order).  u32 keys, whole tiles, plain digits only (probe).
// Measured (2^28 u32): 0.518 ms per pass against 0.503 for rsx_scatter2_kernel on the same box.  The count phase drops
// from 12.2 k to 7.4 k cycles (no load wait left in it), but the 96 KiB of key loads that the other twelve waves have just
// queued sit in front of the digit threads' look-back loads in the CU's memory queue: the chain takes 10.7 k cycles
// instead of 5.6 k and the staging phase waits for it (13.3 k against 8.8 k).  A tile's life stays at 30 k cycles, i.e.
// 8.7 bytes per cycle and CU, which is what reading at 5.5 TB/s plus writing 512-byte runs at 3.1 TB/s
// (tools/ubench/store_runs.hip) add up to: the pass is on its memory-side bound, not on the order of its phases.
#pragma once

#include "rsx_scatter2.hpp"

namespace rsx {

struct Sc5Cfg {
	static constexpr int NWAVES = 16, BLOCK = 1024, KPT = 32, TILE = BLOCK * KPT, LB = 8, SB = 8, VEC = 4, NV = KPT / VEC, CHUNK = 4;
};

struct Sc5Smem {
	__attribute__((aligned(16))) u32 stage[Sc5Cfg::TILE];
	u32 cell[Sc5Cfg::NWAVES][256];
	u32 delta[256];
	u32 wsum[4];
	u32 ticket[2];
};

template <bool TL>
__global__ __launch_bounds__(1024) void rsx_scatter5_kernel(const u32 *__restrict__ kin, u32 *__restrict__ kout, u64 n, u32 shift,
                                                            const u64 *__restrict__ gbase, u32 *status, u32 *ticket, u32 flags, u64 *tl)
{
	typedef Sc5Cfg C;
	typedef StatusBits<u32> SB_;
	constexpr int NWAVES = C::NWAVES, KPT = C::KPT, VEC = C::VEC, NV = C::NV, LB = C::LB, SB = C::SB, CHUNK = C::CHUNK;
	typedef u32 vec_t __attribute__((ext_vector_type(4)));
	__shared__ Sc5Smem sm;
	const u32 tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
	const u32 ntiles = (u32)(n / C::TILE);
	auto opaque = [](u32 x) {
		asm volatile("" : "+v"(x));
		return x;
	};
	u32 *wc = sm.cell[wid];
	u32 *scratch = sm.stage + wid * (64 * KPT);
	const u32 lane_bytes = (wid * (64 * KPT) + lane * VEC) * 4u;
	const __amdgpu_buffer_rsrc_t out_rsrc = __builtin_amdgcn_make_buffer_rsrc(kout, 0, (u32)(n * 4), 0x00020000);
	constexpr u32 TICKET_TID = C::BLOCK - 64;

	if (tid == 0) {
		sm.ticket[0] = atomicAdd(ticket, 1u);
		sm.ticket[1] = atomicAdd(ticket, 1u);
	}
	__syncthreads();
	u32 cur = __builtin_amdgcn_readfirstlane(sm.ticket[0]);
	u32 nxt = __builtin_amdgcn_readfirstlane(sm.ticket[1]);
	if (cur >= ntiles)
		return;
	vec_t v[NV];
	auto load_tile = [&](const u32 tile) {
		const char *tb = (const char *)(kin + (u64)tile * C::TILE);
#pragma unroll
		for (int i = 0; i < NV; ++i)
			v[i] = *(const vec_t *)(tb + i * 1024 + lane_bytes);
	};
	load_tile(cur);
	__syncthreads();   // sm.ticket read by everybody before it is reused

	for (u32 it = 0;; ++it) {
		const u64 t_start = TL ? __builtin_readcyclecounter() : 0;
		u32 keep[KPT];
		u32 tk = 0;
		if (tid == TICKET_TID && nxt < ntiles) {   // the ticket after the next: back long before it is handed over
			typedef __attribute__((address_space(1))) u32 global_u32;
			global_u32 *tp = (global_u32 *)ticket;
			asm volatile("" : "+v"(tp));
			tk = __hip_atomic_fetch_add(tp, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		}
#pragma unroll
		for (int k = 0; k < 4; ++k)
			wc[lane + 64 * k] = 0;
#pragma unroll
		for (int i = 0; i < NV; ++i) {
			*((vec_t *)scratch + i * 64 + lane) = v[i];
			RSX_COMPILER_FENCE();
#pragma unroll
			for (int e = 0; e < VEC; ++e)
				keep[i * VEC + e] = scratch[(i * VEC + e) * 64 + lane];
#pragma unroll
			for (int e = 0; e < VEC; ++e)
				atomicAdd(&wc[(keep[i * VEC + e] >> shift) & 0xFFu], 1u);
			RSX_COMPILER_FENCE();
		}
#pragma unroll
		for (int r = 0; r < KPT; ++r)
			asm volatile("" : "+v"(keep[r]));
		__syncthreads();   // #1
		if (TL && tid == 0)
			tl[(u64)cur * 16 + 1] = __builtin_readcyclecounter();

		u32 tot = 0, incl = 0, tbase = 0;
		u32 w[LB];
		int back = (int)cur - 1;
		u32 *my_status = status + (cur * 256u + tid);
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
				tot += sm.cell[k][tid];
			const u32 word = ((u32)(cur == 0 ? ST_PREFIX : ST_AGGREGATE) << SB_::SHIFT) | tot;
			__hip_atomic_store(my_status, word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			if (cur != 0)
				look();
			u32 x = tot;
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
			tbase = incl - tot;
			for (u32 k = 0; k < wid; ++k)
				tbase += sm.wsum[k];
			u32 acc = tbase;
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
		const bool more = nxt < ntiles;
		if (tid < 256) {
			u64 excl = 0;
			u32 depth = 0;
			if (cur != 0) {
				for (;;) {
					bool done = false;
					int used = 0;
#pragma unroll
					for (int j = 0; j < LB; ++j) {
						const u32 f = w[j] >> SB_::SHIFT;
						if (!done && used == j && f != ST_EMPTY) {
							excl += w[j] & SB_::VALMASK;
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
				__hip_atomic_store(my_status, ((u32)ST_PREFIX << SB_::SHIFT) | (u32)(excl + tot), __ATOMIC_RELAXED,
				                   __HIP_MEMORY_SCOPE_AGENT);
			}
			sm.delta[tid] = (u32)(gbase[tid] + excl - tbase);
			if (TL && tid == 0) {
				tl[(u64)cur * 16 + 3] = __builtin_readcyclecounter();
				tl[(u64)cur * 16 + 12] = depth;
			}
		}
		// the next tile's keys: on their way while this tile is ranked and staged
		if (more)
			load_tile(nxt);
#pragma unroll
		for (int r0 = 0; r0 < KPT; r0 += SB) {
			u32 pos[SB];
#pragma unroll
			for (int r = 0; r < SB; ++r)
				pos[r] = __hip_atomic_fetch_add(&wc[(keep[r0 + r] >> shift) & 0xFFu], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#pragma unroll
			for (int r = 0; r < SB; ++r)
				sm.stage[pos[r]] = keep[r0 + r];
		}
		if (tid == TICKET_TID && more)
			sm.ticket[it & 1] = tk;
		__syncthreads();   // #4
		if (TL && tid == 0)
			tl[(u64)cur * 16 + 4] = __builtin_readcyclecounter();
		const u32 nn = more ? __builtin_amdgcn_readfirstlane(sm.ticket[it & 1]) : ntiles;
#pragma unroll
		for (int j = 0; j < KPT / CHUNK; ++j) {
			if (j % 2 == 0)
				__builtin_amdgcn_sched_barrier(0);
			const u32 i0 = opaque(CHUNK * tid) + CHUNK * j * C::BLOCK;
			const vec_t x = *(const vec_t *)(sm.stage + i0);
			const u32 d0 = (x[0] >> shift) & 0xFFu, d3 = (x[3] >> shift) & 0xFFu;
			if (!(TL && (flags & SCATTER_DBG_NOSTORE))) {
				if (d0 == d3) {
					__builtin_amdgcn_raw_buffer_store_b128(x, out_rsrc, (sm.delta[d0] + i0) * 4u, 0, 0);
				} else {
#pragma unroll
					for (int e = 0; e < CHUNK; ++e)
						__builtin_amdgcn_raw_buffer_store_b32(x[e], out_rsrc, (sm.delta[(x[e] >> shift) & 0xFFu] + i0 + e) * 4u, 0, 0);
				}
			}
		}
		__syncthreads();   // #5
		if (TL && tid == 0)
			tl[(u64)cur * 16 + 5] = __builtin_readcyclecounter();
		if (!more)
			break;
		cur = nxt;
		nxt = nn;
	}
}

}  // namespace rsx
