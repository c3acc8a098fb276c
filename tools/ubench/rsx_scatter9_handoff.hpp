// EXPERIMENT (round 2), not part of the library.  Measured on 2^28 u32 keys (profiles/r02/scatter_probe_v9.txt); output identical
// to rsx_scatter2_kernel's on all four columns, every record read exactly once:
//   HANDOFF = false (this kernel without the hand-off; = rsx_scatter2_kernel's structure):  0.494-0.500 ms, write-out 4.7 k cycles
//   HANDOFF = true:                                                                          0.68-0.70 ms,  write-out 18.6-21 k
//     (records written 1.4 k, whole atoms of the runs 11-13 k, first atoms 5.7-6.7 k; the same wherever the first atoms are
//     written: before, in the middle of or after the runs)
//   timing only, nothing handed over (no record written, none read; the runs' first atoms written partially):  0.580 ms,
//     write-out 10.5 k -- the bookkeeping alone (three more LDS tables, 2 x 4096 lane-tasks per tile) costs twice what partly
//     written atoms cost (2.9 k cycles per tile: DESIGN.md section 4, aligned-run inputs); reading the records adds 1.5 k to
//     the runs' stores (sc1 loads in the same in-order queue), writing them another 0.7 k, waiting for them the rest.
// A tile has 256 runs: every one of them needs a record out and a record in, and a 64-byte record costs as much to move as
// the two partial atoms it saves.  Not pursued.
//
// rsx_scatter9 -- rsx_scatter2_kernel's structure (one-shot workgroup, ticket order, 32 Ki-key tiles, keys only, whole tiles)
// with the RAGGED ENDS OF THE RUNS HANDED TO THE NEXT TILE, so that memory only sees whole 64-byte atoms:
//
//   tile t's run of digit d covers output keys [G, G + c).  With f = G mod A (A = keys per atom):
//     * the keys behind the run's last atom boundary (tau = (G + c) mod A of them) are not written; they are left in a 64-byte
//       record mailbox[t][d] = {A - 1 key slots, tag = READY | tau};
//     * if f > 0 the run starts inside an atom whose first f keys belong to tile t - 1: that tile's record is read, the atom
//       completed with the run's first A - f keys and written with ONE 64-byte store;
//     * a run too short to reach the end of its atom (c < A - f) writes what it has and what it was handed as partial stores
//       and leaves an empty record -- so a tile never depends on anything but its direct predecessor's record, which that
//       tile writes at the start of its write-out from what it staged itself (no chain).
//   A record is written by sixteen lanes of one store instruction (one 64-byte request) and read by sixteen lanes of one load;
//   the reader puts the tag back to 0, the last tile leaves no record.  HANDOFF = false: the same kernel without all this.
#pragma once

#include "rsx_scatter2.hpp"

namespace rsx {

template <typename KT, int LB_ = 8> struct Sc9Cfg {
	static constexpr int NWAVES = 16;
	static constexpr int BLOCK = NWAVES * 64;
	static constexpr int ELEM = sizeof(KT);
	static constexpr int KPT = 128 / ELEM;
	static constexpr int TILE = BLOCK * KPT;
	static constexpr int LB = LB_;
	static constexpr int SB = 8;
	static constexpr int CHUNK = 16 / ELEM;
	static constexpr int STAGE_BYTES = TILE * ELEM;
	static constexpr int ATOM = 64 / ELEM;             // keys per 64-byte atom
	static constexpr u32 RING = 2048;                  // tiles whose records are kept (a record lives until the next tile reads it)
	static_assert(ELEM == 4, "records of fifteen 4-byte keys and a tag");
};

template <typename KT, typename ST, typename C, bool HANDOFF> struct Sc9Smem {
	__attribute__((aligned(16))) unsigned char stage_raw[C::STAGE_BYTES];
	u32 cell[C::NWAVES][256];
	ST delta[256];
	u32 lohi[HANDOFF ? 256 : 1];                        // tile-local bounds (16 bits each) of the part of a run the main write-out stores
	u32 hi[HANDOFF ? 256 : 1];
	u32 rs[HANDOFF ? 256 : 1];                          // tile-local run start
	u32 info[HANDOFF ? 256 : 1];                        // f | tau << 8 | short << 16 | c_small << 20 (c when short)
	u32 wsum[4];
	u32 ticket;
};

enum : u32 { SC9_READY = 0x80000000u, SC9_DBG_NODEPOSIT = 1u << 20, SC9_DBG_PLAINDEPOSIT = 1u << 21, SC9_DBG_NOSPIN = 1u << 22, SC9_DBG_NOREAD = 1u << 23 };

template <typename KT, typename ST, typename C = Sc9Cfg<KT>, bool TL = false, int DIG = DIG_GENERIC, bool HANDOFF = true, int HEAD_AT = 8>
__global__ __launch_bounds__(C::BLOCK) void rsx_scatter9_kernel(const KT *__restrict__ kin, KT *__restrict__ kout, u32 ntiles, u32 shift,
                                                                const u64 *__restrict__ gbase, ST *status, u32 *ticket, u32 *mailbox,
                                                                KdfArgs<KT> ka, u32 flags, u64 *tl)
{
	typedef StatusBits<ST> SB_;
	constexpr int NWAVES = C::NWAVES, BLOCK = C::BLOCK, KPT = C::KPT, SB = C::SB, CHUNK = C::CHUNK, LB = C::LB;
	constexpr u32 A = C::ATOM;
	__shared__ Sc9Smem<KT, ST, C, HANDOFF> sm;
	const u32 tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
	const u64 t_start = TL ? __builtin_readcyclecounter() : 0;
	if (tid == 0)
		sm.ticket = atomicAdd(ticket, 1u);
	for (u32 i = tid; i < NWAVES * 256; i += BLOCK)
		(&sm.cell[0][0])[i] = 0;
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

	// ---- load (element loads: memory order) + count
	KT keep[KPT];
	{
		const KT *p = kin + (u64)tile * C::TILE + (wid * (64 * KPT) + lane);
#pragma unroll
		for (int r = 0; r < KPT; ++r)
			keep[r] = p[r * 64];
#pragma unroll
		for (int r = 0; r < KPT; ++r)
			atomicAdd(&wc[digit2<DIG>(keep[r], ka, shift)], 1u);
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
		const u64 G = gbase[tid] + excl;
		sm.delta[tid] = (ST)(G - tb);
		if constexpr (HANDOFF) {
			const u32 f = (u32)G & (A - 1u), h = (A - f) & (A - 1u);
			const bool shortrun = tc < h;                       // (f > 0 then)
			const u32 tau = (shortrun || tile + 1 == ntiles) ? 0u : ((f + tc) & (A - 1u));   // (nobody behind the last tile)
			sm.rs[tid] = tb;
			const u32 lo_ = shortrun ? tb : tb + h, hi_ = shortrun ? tb : tb + tc - tau;
			sm.lohi[tid] = lo_ | (hi_ << 16);
			sm.hi[tid] = hi_;
			sm.info[tid] = f | (tau << 8) | ((shortrun ? 1u : 0u) << 16) | ((shortrun ? tc : 0u) << 20);
		}
		if (TL && tid == 0) {
			tl[(u64)tile * 16 + 3] = __builtin_readcyclecounter();
			tl[(u64)tile * 16 + 12] = depth;
		}
	}
	// ---- rank + stage
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
	__syncthreads();
	if (TL && tid == 0)
		tl[(u64)tile * 16 + 4] = __builtin_readcyclecounter();

	const ST *delta = sm.delta;
	const bool nostore = TL && (flags & SCATTER_DBG_NOSTORE);
	// ---- hand-off, first half: the records for the next tile (sixteen lanes per digit: lane a holds key slot a, lane 15 the tag)
	if constexpr (HANDOFF) {
		if (tile + 1 < ntiles && !nostore) {
			u32 *myrec = mailbox + (u64)(tile & (C::RING - 1u)) * (256u * 16u);
#pragma unroll
			for (u32 rd = 0; rd < 256u * 16u / BLOCK; ++rd) {
				const u32 d = (tid >> 4) + rd * (BLOCK / 16u), a = tid & 15u;
				const u32 inf = sm.info[d];
				const u32 tau = (inf >> 8) & 0xFFu;
				const bool shortrun = ((inf >> 16) & 1u) != 0;
				if (tau != 0 || shortrun) {
					u32 v = SC9_READY | tau;
					if (a < tau)
						v = (u32)stage_k[sm.hi[d] + a];
					if (flags & SC9_DBG_NODEPOSIT) {
					} else if (flags & SC9_DBG_PLAINDEPOSIT)
						myrec[d * 16u + a] = v;
					else
						__hip_atomic_store(myrec + d * 16u + a, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				}
			}
		}
	}
	if (TL && tid == 0)
		tl[(u64)tile * 16 + 6] = __builtin_readcyclecounter();
	// ---- write-out of the runs' whole atoms (HANDOFF) / of everything, chunks [j0, j1)
	auto main_chunks = [&](const int j0, const int j1) {
#pragma unroll
		for (int j = j0; j < j1; ++j) {
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
				if constexpr (HANDOFF) {
					// every lookup first, no short-circuit: the bounds and offsets of the four elements' runs
					u32 lh[CHUNK];
					ST dl[CHUNK];
					bool in[CHUNK];
#pragma unroll
					for (int e = 0; e < CHUNK; ++e) {
						lh[e] = sm.lohi[d[e]];
						dl[e] = delta[d[e]];
					}
#pragma unroll
					for (int e = 0; e < CHUNK; ++e)
						in[e] = (i0 + e >= (lh[e] & 0xFFFFu)) & (i0 + e < (lh[e] >> 16));
					if ((d[0] == d[CHUNK - 1]) & in[0] & in[CHUNK - 1]) {
						store_chunk<KT, CHUNK>(kout + (ST)(dl[0] + i0), kv);
					} else {
#pragma unroll
						for (int e = 0; e < CHUNK; ++e)
							if (in[e])
								kout[(ST)(dl[e] + i0 + e)] = kv[e];
					}
				} else {
					if (d[0] == d[CHUNK - 1]) {
						store_chunk<KT, CHUNK>(kout + (ST)(delta[d[0]] + i0), kv);
					} else {
#pragma unroll
						for (int e = 0; e < CHUNK; ++e)
							kout[(ST)(delta[d[e]] + i0 + e)] = kv[e];
					}
				}
			}
		}
	};
	constexpr u32 NRD = 256u * 16u / BLOCK;
	u32 rec[HANDOFF ? NRD : 1];
	u32 *prec = mailbox + (u64)((tile - 1u) & (C::RING - 1u)) * (256u * 16u);
	if constexpr (HANDOFF) {
		// the previous tile's records, all four rounds requested at once (whether needed or not: no branch, no wait)
#pragma unroll
		for (u32 rd = 0; rd < NRD; ++rd)
			rec[rd] = (flags & SC9_DBG_NOREAD) ? SC9_READY : __hip_atomic_load(prec + ((tid >> 4) + rd * (BLOCK / 16u)) * 16u + (tid & 15u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	}
	main_chunks(0, HEAD_AT);
	if (TL && tid == 0)
		tl[(u64)tile * 16 + 7] = __builtin_readcyclecounter();
	// ---- hand-off, second half: the atoms the runs start in, completed with the previous tile's records
	if constexpr (HANDOFF) {
		if (!nostore) {
			const u32 a = tid & 15u;
#pragma unroll
			for (u32 rd = 0; rd < NRD; ++rd) {
				const u32 d = (tid >> 4) + rd * (BLOCK / 16u);
				const u32 inf = sm.info[d];
				const u32 f = inf & 0xFFu;
				const bool shortrun = ((inf >> 16) & 1u) != 0;
				const u32 cs = inf >> 20;
				if (f != 0) {
					u32 len = 0;
					if (tile != 0) {
						u32 tag = __shfl(rec[rd], 15, 16), spins = 0;
						while (!(tag & SC9_READY) && !(flags & SC9_DBG_NOSPIN) && ++spins < (1u << 22)) {   // (probe: bounded)
							rec[rd] = __hip_atomic_load(prec + d * 16u + a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
							tag = __shfl(rec[rd], 15, 16);
						}
						len = tag & 0xFFu;
						if (a == 15u && !(flags & SC9_DBG_NOSPIN))
							__hip_atomic_store(prec + d * 16u + 15u, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
					}
					// the atom's key slot a: handed over (a < f, if the previous tile left them) or the run's own (a - f) -th key
					const u32 own = shortrun ? cs : A - f;   // keys of this run in the atom
					const ST at = (ST)(delta[d] + sm.rs[d]) - (ST)f;   // output index of the atom's first key
					if (a < f) {
						if (len != 0)
							kout[(ST)(at + a)] = (KT)rec[rd];
					} else if (a - f < own) {
						kout[(ST)(at + a)] = stage_k[sm.rs[d] + (a - f)];
					}
				}
			}
		}
	}
	if (TL && tid == 0)
		tl[(u64)tile * 16 + 8] = __builtin_readcyclecounter();
	main_chunks(HEAD_AT, KPT / CHUNK);
	if (TL && tid == 0)
		tl[(u64)tile * 16 + 5] = __builtin_readcyclecounter();
}

}  // namespace rsx
