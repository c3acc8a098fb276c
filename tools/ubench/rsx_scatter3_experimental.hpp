// rsx_scatter3_experimental.hpp -- EXPERIMENT, not part of the library: a persistent, software-pipelined scatter
// pass for keys without a payload.  Measured on MI355X (2^28 u32, tools/ubench/scatter_probe.hip): 0.64 ms per pass
// against 0.53-0.56 ms for rsx_scatter2_kernel, bit-exact.  Why it does not win is in DESIGN.md ("what was tried"):
// the pass is bound by how HBM takes 512-byte runs that start and end inside 64-byte atoms (about 3 TB/s for the
// stores), not by the overlap of loads, stores and LDS work this schedule buys.
//
// Same job as rsx_scatter2_kernel (rsx_scatter2.hpp): one stable scatter pass of the reference's loop at
// radix_sort.hpp:82-90, count first, returning LDS atomics as ranks, 32 Ki-key tiles staged through LDS.
// What is different is the schedule.  A workgroup is persistent (one per CU: 128 KiB of staging is all a CU
// holds) and takes tile after tile by atomic ticket (= chain order).  Per tile a_i:
//
//   F(i)    fused, per wave and without a workgroup barrier: the wave writes OUT its own 8 KiB region of the
//           staged tile a_(i-1), 1 KiB per step, and refills each KiB at once with the keys of a_i, which it
//           loaded into registers a tile ago (16 bytes per lane, read back in rank order = the transposition);
//           the freed registers are reloaded with the keys of a_(i+1); the refilled keys are counted into the
//           wave's row of (wave, digit) cells.  Stores, loads and LDS atomics of different steps overlap.
//   mid(i)  barrier; the digit threads (waves 0-3) total the cells, publish the tile's aggregate, start the
//           look-back and scan; barrier; cells become run starts; barrier; every wave stages its keys (one
//           returning LDS atomic per key = position), waves 4-15 at once, the digit threads after they have
//           resolved the chain and published the inclusive prefix; barrier.
//
// A CU's vector-memory path is one queue shared by its waves, and per wave vector-memory operations complete
// in order: the keys of the next tile are requested a whole tile ahead and consumed one tile later, so that
// neither the look-back's loads nor the ticket wait behind bulk traffic they did not have to.
//
// Stability rests on the same property as rsx_scatter2_kernel (returning LDS atomics resolve same-address
// lanes in lane order); the host selects this kernel only after the device self-check (rsx.hip).
#pragma once

#include "rsx_scatter2.hpp"

namespace rsx {

template <typename KT, int NWAVES_ = 16, int LB_ = 8, int STAGGER_ = 30000> struct Sc3Cfg {
	static constexpr int STAGGER = STAGGER_;          // shader-clock cycles the workgroups' starts are spread over (about one tile period)
	static constexpr int NWAVES = NWAVES_;
	static constexpr int BLOCK = NWAVES * 64;
	static constexpr int ELEM = sizeof(KT);
	static constexpr int KPT = ELEM == 8 ? 16 : 32;    // keys per lane: 128 KiB of staging at 16 waves of 4/8-byte keys
	static constexpr int TILE = BLOCK * KPT;
	static constexpr int LB = LB_;                     // status words fetched per look-back round trip
	static constexpr int SB = 8;                       // returning atomics a lane has in flight while staging
	static constexpr int VEC = 16 / ELEM;              // elements per 16-byte load / staged chunk a lane writes out
	static constexpr int NV = KPT / VEC;               // 16-byte loads per lane and tile = steps of F
	static constexpr int STAGE_BYTES = TILE * ELEM;
	static_assert(NWAVES >= 4, "256 digit threads are needed");
	static_assert(KPT % SB == 0 && KPT % VEC == 0, "whole batches / vectors per lane");
};

template <typename KT, typename ST, typename C> struct Sc3Smem {
	__attribute__((aligned(16))) unsigned char stage_raw[C::STAGE_BYTES];
	u32 cell[C::NWAVES][256];   // per (wave, digit): count, then run start / cursor
	ST delta[256];              // global offset of a digit's run minus its tile-local offset
	u32 wsum[4];
	u32 ticket;                 // the tile after the next
	u32 first[2];
};

// BUF: the output is smaller than 4 GiB and is written through a buffer resource (see write_chunk).
template <typename KT, typename ST, typename C = Sc3Cfg<KT>, bool TL = false, int DIG = DIG_GENERIC, bool BUF = true>
__global__ __launch_bounds__(C::BLOCK) void rsx_scatter3_kernel(const KT *__restrict__ kin, KT *__restrict__ kout, u64 n, u32 shift,
                                                                 const u64 *__restrict__ gbase, ST *status, u32 *ticket,
                                                                 KdfArgs<KT> ka, u32 flags, const uint8_t *__restrict__ lut, u64 *tl)
{
	typedef StatusBits<ST> SB_;
	constexpr int NWAVES = C::NWAVES, BLOCK = C::BLOCK, KPT = C::KPT, SB = C::SB, LB = C::LB;
	constexpr int VEC = C::VEC, NV = C::NV;
	constexpr bool WIDE_STORE = sizeof(KT) >= 4;   // 16-byte stores of VEC keys (needs 4-byte alignment only)
	typedef KT vec_t __attribute__((ext_vector_type(VEC)));
	typedef vec_t uvec_t __attribute__((aligned(sizeof(KT))));   // the caller's keys are only element-aligned
	__shared__ Sc3Smem<KT, ST, C> sm;
	const u32 tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
	const u32 ntiles = (u32)((n + C::TILE - 1) / C::TILE);
	const u32 nfull = (u32)(n / C::TILE);           // tiles [0, nfull) are whole
	KT *const stage_k = (KT *)sm.stage_raw;
	KT *const slice = stage_k + wid * (64 * KPT);   // wave w owns [w*64*KPT, +64*KPT) of a tile
	const u32 wofs = wid * (64 * KPT) + lane;       // round r of a lane: element wofs + 64 r
	u32 *const wc = sm.cell[wid];
	const __amdgpu_buffer_rsrc_t out_rsrc = __builtin_amdgcn_make_buffer_rsrc(kout, 0, BUF ? (u32)(n * sizeof(KT)) : 0u, 0x00020000);

	if (tid == 0) {
		// tiles are handed out in start order => look-back cannot deadlock.  A workgroup holds the tickets of its
		// current tile, of the next (whose keys are in flight) and, from the middle of a tile on, of the one after.
		const u32 a0 = atomicAdd(ticket, 1u);
		// Workgroups that start together stay together: every CU would load and store in the same half of the
		// tile period and leave HBM idle in the other.  So the start is staggered over one tile period by first
		// ticket, before the second ticket is taken (tickets must be processed in the order they are taken).
		if (C::STAGGER != 0 && a0 < gridDim.x) {
			const u64 until = __builtin_readcyclecounter() + (u64)a0 * C::STAGGER / gridDim.x;
			while (__builtin_readcyclecounter() < until)
				__builtin_amdgcn_s_sleep(8);
		}
		sm.first[0] = a0;
		sm.first[1] = atomicAdd(ticket, 1u);
	}
	__syncthreads();
	u32 cur = __builtin_amdgcn_readfirstlane(sm.first[0]);   // a_i
	u32 nxt = __builtin_amdgcn_readfirstlane(sm.first[1]);   // a_(i+1)
	if (cur >= ntiles)
		return;

	vec_t v[NV];      // lane l: elements VEC (64 j + l) .. +VEC-1 of its wave's slice of the tile in flight, j < NV
	KT keep[KPT];     // the current tile's keys in rank order: round r = element 64 r + lane of the slice
	// (uniform base + one 32-bit lane offset: a scalar address and a single offset register for all NV loads)
	const u32 lane_bytes = (wid * (64 * KPT) + lane * VEC) * (u32)sizeof(KT);
	auto load_vec = [&](const int j, const u32 tile) {
		const char *tb = (const char *)(kin + (u64)tile * C::TILE) + j * 1024;
		v[j] = *(const uvec_t *)(tb + lane_bytes);
	};
	// Elements past the end of a partial tile are counted and staged as digit 255: they are last in memory order,
	// so they land behind every real element of the tile (positions >= cnt) and are never written out.
	auto dig = [&](auto full_c, const int r, const u32 cnt) -> u32 {
		const u32 d = digit2<DIG>(keep[r], ka, shift, flags, lut);
		if constexpr (decltype(full_c)::value)
			return d;
		else
			return wofs + r * 64 < cnt ? d : 255u;
	};
	auto zero_row = [&]() {
#pragma unroll
		for (int k = 0; k < 4; ++k)
			wc[lane + 64 * k] = 0;
	};
	// One staged chunk (VEC consecutive elements at tile-local index i0) to its place.  Consecutive staged elements of
	// one digit go to consecutive addresses: when the chunk lies in one run (first digit == last) it is one wide store,
	// chunks straddling a run boundary go element-wise.
	auto write_chunk = [&](auto full_c, const u32 i0, const u32 cnt) {
		constexpr bool full = decltype(full_c)::value;
		const vec_t x = *(const vec_t *)(stage_k + i0);
		KT kv[VEC];
		u32 d[VEC];
#pragma unroll
		for (int e = 0; e < VEC; ++e) {
			kv[e] = x[e];
			d[e] = digit2<DIG>(kv[e], ka, shift, flags, lut);
		}
		if (TL && (flags & SCATTER_DBG_NOSTORE))
			return;
		const bool whole = full || i0 + VEC <= cnt;
		const bool one_run = WIDE_STORE && whole && d[0] == d[VEC - 1];
		if constexpr (BUF && WIDE_STORE) {
			// unconditional: lanes whose chunk is not one run store out of the buffer's range, which the hardware
			// drops.  (A store every lane executes is one the compiler can count when it waits for loads issued
			// before it; behind a branch it would have to let all stores drain.)
			const u32 off = one_run ? (u32)(sm.delta[d[0]] + i0) * (u32)sizeof(KT) : 0xFFFFFFF0u;
			u32x4 raw;
			__builtin_memcpy(&raw, &x, 16);
			__builtin_amdgcn_raw_buffer_store_b128(raw, out_rsrc, off, 0, 0);
		}
		if (one_run) {
			if constexpr (!(BUF && WIDE_STORE))
				store_chunk<KT, VEC>(kout + (ST)(sm.delta[d[0]] + i0), kv);
		} else {
#pragma unroll
			for (int e = 0; e < VEC; ++e)
				if (full || i0 + e < cnt) {
					if constexpr (BUF && sizeof(KT) == 4)
						__builtin_amdgcn_raw_buffer_store_b32((u32)kv[e], out_rsrc, (u32)(sm.delta[d[e]] + i0 + e) * 4u, 0, 0);
					else
						kout[(ST)(sm.delta[d[e]] + i0 + e)] = kv[e];
				}
		}
	};
	// the write-out alone (the workgroup's last tile, or before a partial tile)
	// (the chunk index is recomputed per step from an opaque copy: hoisted out of the tile loop, the NV indices and
	// what derives from them would take registers from the keys in flight)
	auto chunk_index = [&](const int j) -> u32 {
		u32 b = wid * (64 * KPT) + VEC * lane;
		asm volatile("" : "+v"(b));
		return b + VEC * 64 * j;
	};
	auto write_out = [&](auto full_c, const u32 cnt) __attribute__((always_inline)) {
#pragma unroll
		for (int j = 0; j < NV; ++j)
			write_chunk(full_c, chunk_index(j), cnt);
	};
	// F: write out the staged tile (if WRITE), refill with the keys in v, reload v with tile `pre` (if it is a whole
	// tile), read the refilled keys back in rank order and count them
	auto fused = [&](auto write_c, const u32 pre) __attribute__((always_inline)) {
		constexpr bool WRITE = decltype(write_c)::value;
		const bool reload = pre < nfull;
		zero_row();
#pragma unroll
		for (int j = 0; j < NV; ++j) {
			__builtin_amdgcn_sched_barrier(0);   // one step's registers at a time
			if constexpr (WRITE)
				write_chunk(std::true_type{}, chunk_index(j), (u32)C::TILE);
			*((vec_t *)slice + j * 64 + lane) = v[j];   // DS operations of one wave execute in issue order
			if (reload)
				load_vec(j, pre);
			RSX_COMPILER_FENCE();
#pragma unroll
			for (int r = j * VEC; r < (j + 1) * VEC; ++r)
				keep[r] = slice[r * 64 + lane];
#pragma unroll
			for (int r = j * VEC; r < (j + 1) * VEC; ++r)
				atomicAdd(&wc[dig(std::true_type{}, r, 0u)], 1u);
		}
	};
	// the last, partial tile: element-wise, not prefetched
	auto fill_partial = [&](const u64 base, const u32 cnt) __attribute__((always_inline)) {
		zero_row();
#pragma unroll 4
		for (int r = 0; r < KPT; ++r) {
			const u32 o = wofs + r * 64;
			slice[r * 64 + lane] = o < cnt ? kin[base + o] : (KT)0;
		}
		RSX_COMPILER_FENCE();
#pragma unroll
		for (int r = 0; r < KPT; ++r)
			keep[r] = slice[r * 64 + lane];
#pragma unroll
		for (int r = 0; r < KPT; ++r)
			atomicAdd(&wc[dig(std::false_type{}, r, cnt)], 1u);
	};

	// The ticket for a_(i+2) is taken in the middle of tile a_i by one thread of a wave without digit threads (those
	// are the critical path).  (The pointer is made opaque so that the compiler issues one plain returning atomic
	// instead of its wave-aggregated form, which waits for the result at once.)
	constexpr u32 TICKET_TID = BLOCK - 64;
	u32 tk = 0;
	auto take_ticket = [&]() {
		if (tid == TICKET_TID) {
			typedef __attribute__((address_space(1))) u32 global_u32;
			global_u32 *tp = (global_u32 *)ticket;
			asm volatile("" : "+v"(tp));
			tk = __hip_atomic_fetch_add(tp, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		}
	};

	// mid: from the counted tile to the staged tile
	auto mid = [&](auto full_c, const u32 cnt, const u64 t_start) __attribute__((always_inline)) {
		constexpr bool full = decltype(full_c)::value;
		// the cell addresses are recomputed when the keys are staged (two VALU operations each) rather than kept
		// in 32 registers across the chain
#pragma unroll
		for (int r = 0; r < KPT; ++r)
			asm volatile("" : "+v"(keep[r]));
		__syncthreads();   // #1: counts complete; everybody is done with the previous tile (delta, staging area)
		if (TL && tid == 0) {
			tl[(u64)cur * 16 + 0] = t_start;
			tl[(u64)cur * 16 + 1] = __builtin_readcyclecounter();
			tl[(u64)cur * 16 + 9] = __builtin_amdgcn_s_memrealtime();   // 100 MHz, the same clock on every XCD
		}

		// ---- digit thread d: tile total, publish the aggregate, start the look-back, scan over digits
		u64 gb = 0;
		u32 tot = 0, own = 0, incl = 0, tbase = 0;
		ST w[LB];
		int back = (int)cur - 1;   // nearest predecessor not consumed yet
		ST *my_status = status + (cur * 256u + tid);   // (32-bit element offsets from the uniform base)
		auto look = [&]() {
			u32 t = tid;
			asm volatile("" : "+v"(t));   // (or the compiler keeps LB loop-invariant offsets in registers)
#pragma unroll
			for (int j = 0; j < LB; ++j) {
				const int p = back - j > 0 ? back - j : 0;   // tile 0 always holds a prefix: safe filler
				w[j] = __hip_atomic_load(status + ((u32)p * 256u + t), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			}
		};
		if (tid < 256) {
#pragma unroll
			for (int k = 0; k < NWAVES; ++k)
				tot += sm.cell[k][tid];
			own = tot;   // what the tile really holds of this digit
			if (!full && tid == 255)
				own -= (u32)C::TILE - cnt;
			const ST word = ((ST)(cur == 0 ? ST_PREFIX : ST_AGGREGATE) << SB_::SHIFT) | (ST)own;
			__hip_atomic_store(my_status, word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			if (cur != 0)
				look();
			gb = gbase[tid];   // (per tile rather than kept in two registers)
			u32 x = tot;
#pragma unroll
			for (int off = 1; off < 64; off <<= 1) {
				const u32 y = __shfl_up(x, off);
				if (lane >= (u32)off)
					x += y;
			}
			incl = x;
			if (lane == 63) {
				u32 wv = wid;
				asm volatile("" : "+v"(wv));   // (address computed here, not kept in a register across the tile)
				sm.wsum[wv] = x;
			}
		}
		if (full && nxt < ntiles)
			take_ticket();
		__syncthreads();   // #2: wave totals of the digit scan
		if (tid < 256) {
			tbase = incl - tot;
			for (u32 k = 0; k < wid; ++k)
				tbase += sm.wsum[k];
			u32 acc = tbase;   // counts -> run starts
#pragma unroll
			for (int k = 0; k < NWAVES; ++k) {
				const u32 c = sm.cell[k][tid];   // (read again rather than kept: the registers hold the look-back window)
				sm.cell[k][tid] = acc;
				acc += c;
			}
		}
		__syncthreads();   // #3: cursors in place
		if (TL && tid == 0)
			tl[(u64)cur * 16 + 2] = __builtin_readcyclecounter();

		// ---- waves 4..15 stage at once; the digit threads first resolve the chain
		if (tid < 256) {
			u64 excl = 0;
			u32 depth = 0;
			if (cur != 0) {
				// LB predecessors per round trip (independent loads), consumed in order: aggregates are summed
				// until the first inclusive prefix; an empty word ends the batch.
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
				const ST pword = ((ST)ST_PREFIX << SB_::SHIFT) | (ST)(excl + own);
				__hip_atomic_store(my_status, pword, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			}
			sm.delta[tid] = (ST)(gb + excl - tbase);   // modulo 2^32 when ST is 32-bit (n < 2^30 then)
			if (TL && tid == 0) {
				tl[(u64)cur * 16 + 3] = __builtin_readcyclecounter();
				tl[(u64)cur * 16 + 12] = depth;
			}
		}
		// rank + stage: the returning atomic on the (wave, digit) cursor is the key's tile-local position.  Rounds are
		// issued in memory order; lanes of a round come back in lane order.  All the batch's atomics are issued before
		// the first position is needed, then the keys are stored at their positions.
#pragma unroll
		for (int r0 = 0; r0 < KPT; r0 += SB) {
			u32 pos[SB];
#pragma unroll
			for (int r = 0; r < SB; ++r)
				pos[r] = __hip_atomic_fetch_add(&wc[dig(full_c, r0 + r, cnt)], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#pragma unroll
			for (int r = 0; r < SB; ++r)
				stage_k[pos[r]] = keep[r0 + r];
		}
		if (full && nxt < ntiles && tid == TICKET_TID)
			sm.ticket = tk;
		__syncthreads();   // #4: tile staged, delta in place
		if (TL && tid == 0) {
			tl[(u64)cur * 16 + 4] = __builtin_readcyclecounter();
			tl[(u64)cur * 16 + 10] = __builtin_amdgcn_s_memrealtime();
		}
	};

	// ---- the first tile: nothing to write out
	u64 t0 = TL ? __builtin_readcyclecounter() : 0;
	if (cur < nfull) {
#pragma unroll
		for (int j = 0; j < NV; ++j)
			load_vec(j, cur);
		fused(std::false_type{}, nxt);
	} else {
		fill_partial((u64)cur * C::TILE, (u32)(n - (u64)cur * C::TILE));
	}
	for (;;) {
		if (cur < nfull) {
			mid(std::true_type{}, (u32)C::TILE, t0);
		} else {
			// the partial tile is the last one in ticket order: whoever gets it ends with it
			const u32 cnt = (u32)(n - (u64)cur * C::TILE);
			mid(std::false_type{}, cnt, t0);
			write_out(std::false_type{}, cnt);
			break;
		}
		if (nxt >= ntiles) {
			write_out(std::true_type{}, (u32)C::TILE);
			break;
		}
		const u32 nn = __builtin_amdgcn_readfirstlane(sm.ticket);   // a_(i+2)
		t0 = TL ? __builtin_readcyclecounter() : 0;
		if (nxt < nfull) {
			fused(std::true_type{}, nn);
		} else {
			write_out(std::true_type{}, (u32)C::TILE);
			fill_partial((u64)nxt * C::TILE, (u32)(n - (u64)nxt * C::TILE));
		}
		cur = nxt;
		nxt = nn;
	}
}

}  // namespace rsx
