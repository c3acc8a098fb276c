// rsx_scatter4_experimental.hpp -- EXPERIMENT, not part of the library: a scatter pass whose tile is TWICE the staging
// area (64 Ki four-byte keys per workgroup, staged and written out in two windows of 32 Ki positions), so that a digit's
// run is 1 KiB instead of 512 bytes (tools/ubench/store_runs.hip: 3.8 against 3.0 TB/s for runs that start anywhere).
//
// The keys of the tile stay in registers (64 per lane).  Counting is as in rsx_scatter2_kernel; the ranking atomics are
// run once per window from a pristine copy of the run starts (16-bit, tile-local positions < 65536) and a key is staged in
// the window its position falls into.  Whole tiles only, unsigned ascending keys only (probe).
// Measured (2^28 u32): 0.76 ms per pass against 0.50 -- three LDS atomics per key instead of two, 64 + 8 registers of keys
// and addresses spilling, and a tile life of 117 k cycles against 2 x 29 k.
#pragma once

#include "rsx_scatter2.hpp"

namespace rsx {

struct Sc4Cfg {
	static constexpr int NWAVES = 16, BLOCK = 1024, KPT = 64, TILE = BLOCK * KPT, WIN = TILE / 2, LB = 4;
	static constexpr int VEC = 4, NV = KPT / VEC, CHUNK = 4;
};

struct Sc4Smem {
	__attribute__((aligned(16))) u32 stage[Sc4Cfg::WIN];   // 128 KiB
	u32 cell[Sc4Cfg::NWAVES][256];                         // counts, then cursors
	unsigned short start[Sc4Cfg::NWAVES][256];             // pristine run starts (tile-local)
	u32 delta[256];
	u32 wsum[4];
	u32 ticket;
};

template <bool TL>
__global__ __launch_bounds__(1024) void rsx_scatter4_kernel(const u32 *__restrict__ kin, u32 *__restrict__ kout, u64 n, u32 shift,
                                                            const u64 *__restrict__ gbase, u32 *status, u32 *ticket, u32 flags, u64 *tl)
{
	typedef Sc4Cfg C;
	typedef StatusBits<u32> SB_;
	constexpr int NWAVES = C::NWAVES, KPT = C::KPT, VEC = C::VEC, NV = C::NV, LB = C::LB, CHUNK = C::CHUNK;
	typedef u32 vec_t __attribute__((ext_vector_type(4)));
	__shared__ Sc4Smem sm;
	const u32 tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
	const u64 t_start = TL ? __builtin_readcyclecounter() : 0;
	auto opaque = [](u32 x) {
		asm volatile("" : "+v"(x));
		return x;
	};
	if (tid == 0)
		sm.ticket = atomicAdd(ticket, 1u);
	for (u32 i = tid; i < NWAVES * 256; i += C::BLOCK)
		(&sm.cell[0][0])[i] = 0;
	__syncthreads();
	const u32 tile = __builtin_amdgcn_readfirstlane(sm.ticket);
	const u64 base = (u64)tile * C::TILE;
	u32 *wc = sm.cell[wid];
	u32 keep[KPT];
	const __amdgpu_buffer_rsrc_t out_rsrc = __builtin_amdgcn_make_buffer_rsrc(kout, 0, (u32)(n * 4), 0x00020000);

	// ---- load + count, in two halves of 32 rounds (the wave's slice of the staging area holds one half: 8 KiB)
	{
		const u32 lane_bytes = (wid * (64 * KPT) + lane * VEC) * 4u;   // uniform base + one 32-bit lane offset
		u32 *scratch = sm.stage + wid * (64 * KPT / 2);
		vec_t v[NV / 2];
#pragma unroll
		for (int h = 0; h < 2; ++h) {
#pragma unroll
			for (int i = 0; i < NV / 2; ++i)
				v[i] = *(const vec_t *)((const char *)(kin + base) + (h * (NV / 2) + i) * 1024 + lane_bytes);
#pragma unroll
			for (int i = 0; i < NV / 2; ++i) {
				*((vec_t *)scratch + i * 64 + lane) = v[i];
				RSX_COMPILER_FENCE();
#pragma unroll
				for (int e = 0; e < VEC; ++e)
					keep[h * (KPT / 2) + i * VEC + e] = scratch[(i * VEC + e) * 64 + lane];
#pragma unroll
				for (int e = 0; e < VEC; ++e)
					atomicAdd(&wc[(keep[h * (KPT / 2) + i * VEC + e] >> shift) & 0xFFu], 1u);
				RSX_COMPILER_FENCE();
			}
		}
	}
#pragma unroll
	for (int r = 0; r < KPT; ++r)
		asm volatile("" : "+v"(keep[r]));
	__syncthreads();
	if (TL && tid == 0)
		tl[(u64)tile * 16 + 1] = __builtin_readcyclecounter();

	// ---- digit threads: totals, aggregate, look-back start, scan
	u32 tot = 0, incl = 0, tbase = 0;
	u32 w[LB];
	int back = (int)tile - 1;
	u32 *my_status = status + (tile * 256u + tid);
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
		const u32 word = ((u32)(tile == 0 ? ST_PREFIX : ST_AGGREGATE) << SB_::SHIFT) | tot;
		__hip_atomic_store(my_status, word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		if (tile != 0)
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
	__syncthreads();
	if (tid < 256) {
		tbase = incl - tot;
		for (u32 k = 0; k < wid; ++k)
			tbase += sm.wsum[k];
		u32 acc = tbase;
#pragma unroll
		for (int k = 0; k < NWAVES; ++k) {
			const u32 c = sm.cell[k][tid];
			sm.start[k][tid] = (unsigned short)acc;
			acc += c;
		}
	}
	__syncthreads();
	if (TL && tid == 0) {
		tl[(u64)tile * 16 + 0] = t_start;
		tl[(u64)tile * 16 + 2] = __builtin_readcyclecounter();
	}
	bool chain_done = false;
#pragma unroll 1
	for (u32 win = 0; win < 2; ++win) {
		// cursors from the pristine run starts (own row)
#pragma unroll
		for (int k = 0; k < 4; ++k)
			wc[lane + 64 * k] = sm.start[wid][lane + 64 * k];
		if (tid < 256 && !chain_done) {
			u64 excl = 0;
			u32 depth = 0;
			if (tile != 0) {
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
				tl[(u64)tile * 16 + 3] = __builtin_readcyclecounter();
				tl[(u64)tile * 16 + 12] = depth;
			}
		}
		chain_done = true;
		// rank every key again, stage the ones whose position lies in this window
#pragma unroll
		for (int r0 = 0; r0 < KPT; r0 += 8) {
			__builtin_amdgcn_sched_barrier(0);
			u32 pos[8];
#pragma unroll
			for (int r = 0; r < 8; ++r)
				pos[r] = __hip_atomic_fetch_add(&wc[(keep[r0 + r] >> shift) & 0xFFu], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#pragma unroll
			for (int r = 0; r < 8; ++r)
				if ((pos[r] >> 15) == win)
					sm.stage[pos[r] & (C::WIN - 1)] = keep[r0 + r];
		}
		__syncthreads();
		if (TL && tid == 0)
			tl[(u64)tile * 16 + 4 + 2 * win] = __builtin_readcyclecounter();
#pragma unroll
		for (int j = 0; j < C::WIN / (C::BLOCK * CHUNK); ++j) {
			if (j % 2 == 0)
				__builtin_amdgcn_sched_barrier(0);
			const u32 i0 = opaque(CHUNK * tid) + CHUNK * j * C::BLOCK;
			const vec_t x = *(const vec_t *)(sm.stage + i0);
			u32 kv[CHUNK], d[CHUNK];
#pragma unroll
			for (int e = 0; e < CHUNK; ++e) {
				kv[e] = x[e];
				d[e] = (kv[e] >> shift) & 0xFFu;
			}
			if (!(TL && (flags & SCATTER_DBG_NOSTORE))) {
				const u32 g0 = win * C::WIN + i0;
				if (d[0] == d[CHUNK - 1]) {
					u32x4 raw = {kv[0], kv[1], kv[2], kv[3]};
					__builtin_amdgcn_raw_buffer_store_b128(raw, out_rsrc, (sm.delta[d[0]] + g0) * 4u, 0, 0);
				} else {
#pragma unroll
					for (int e = 0; e < CHUNK; ++e)
						__builtin_amdgcn_raw_buffer_store_b32(kv[e], out_rsrc, (sm.delta[d[e]] + g0 + e) * 4u, 0, 0);
				}
			}
		}
		__syncthreads();
		if (TL && tid == 0)
			tl[(u64)tile * 16 + 5 + 2 * win] = __builtin_readcyclecounter();
	}
}

}  // namespace rsx
