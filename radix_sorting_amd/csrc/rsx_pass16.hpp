// rsx_pass16.hpp -- the LEVEL-2 pass of a keys-only sort without a histogram, 4-byte keys into two-byte slots (round 5), gfx950.
//
// What the pass has to do (rsx_hybrid.hpp): the level-1 pass has left 256 buckets in slots; every key of bucket b goes
// to slot (b, d) of the scratch array, d = its level-2 digit, as the low sixteen bits of its DERIVED key -- the leaves
// (rsx_leaf16.hpp) sort those and put the rest back from the slot's number.  The reference's loop makes the same trip per kept
// column (radix_sort.hpp:82-90); the order INSIDE a slot is free here (the leaves emit any ascending order of equal bits).
//
// Through round 4 this was an instantiation of rsx_scatter2_kernel (KTO = u16, SEG, nine template parameters and a prologue of
// ten run-time branches).  It staged whole 4-byte keys, narrowed them in the write-out and so issued as many store
// instructions as the level-1 pass for half the bytes (8 bytes per lane), chained its tiles by decoupled look-back although
// nothing needs their order, and held one 128 KiB tile per CU: 496 us for 1.5 GiB = 0.41 of the HBM peak, the kernel furthest
// below its roofline in BASELINE.json's headline (profiles/r04/bench/roofline_table.json).  This kernel is that pass and
// nothing else:
//
//   * the tile's VALUES are staged (two bytes each): 64 KiB of LDS per 32 Ki-key tile, so TWO workgroups are resident per CU and
//     one tile's loads overlap the other's ranking and stores -- the overlap that every 128 KiB design lacked;
//   * no chain: a digit's place in its slot comes from ONE returning global atomic per (tile, digit) on the slot's cursor -- the
//     order of tiles inside a slot is arbitrary, nothing waits for a predecessor, a second resident workgroup cannot delay
//     anybody's look-back (finding 4 of DESIGN.md 4); the cursors are the status words of each bucket's LAST tile, where
//     rsx_seg_slack_plan_kernel reads the (digit, digit) counts anyway;
//   * no rank registers and no reliance on the lane order of returning LDS atomics: count with a plain atomic, place with a
//     returning one on the run's cursor (any order inside a run will do) -- 32 key registers per lane, 64 in all;
//   * the keys are read with 16-byte loads and never transposed (the order inside a tile is free too);
//   * the write-out moves 16 bytes per lane (eight values): which run a group of eight staged values lies in comes from a
//     4096-entry table the digit threads fill while the other waves stage; groups that straddle a run boundary (one in
//     sixteen) go value by value.
#pragma once

#include "rsx_kernels.hpp"
#include "rsx_hybrid.hpp"

namespace rsx {

// WGS: workgroups per CU the register allocation leaves room for (1: 128 registers per lane, 2: 64)
template <int WGS_ = 2> struct Pass16Cfg {
	static constexpr int WGS = WGS_;
	static constexpr int BLOCK = 1024, KPT = 32, TILE = BLOCK * KPT;   // the tile of Sc2Cfg<u32, NoVal>: the tile table is shared
	static constexpr int NGROUP = TILE / 8;                            // groups of eight staged values (16 bytes)
};

template <int WGS> struct Pass16Smem {
	__attribute__((aligned(16))) unsigned short stage[32768 + (WGS == 1 ? 8192 : 0)];   // (WGS == 1, the probe: too large for two per CU)   // the tile's values in slot order (swizzled, rsx_scatter2.hpp)
	u32 cell[256];              // per digit: count, then the run's cursor (tile-local)
	u32 delta[256];             // slot position of the run's first value minus its tile-local position
	unsigned short rend[256];   // tile-local end of the digit's run
	unsigned char group_digit[4096];   // the digit of the first value of every group of eight
	u32 wsum[4];
};

// kin / kin_hi / lo_slots: the level-1 slots lie in two arrays (SegArgs, rsx_scatter2.hpp): buckets below lo_slots in kin, the
// others at the same element index of kin_hi (kin_hi == nullptr: all in kin).  cursors: the pass's status region (zeroed by
// rsx_blind_precheck_kernel), 256 words per tile of the table; btile[b + 1] - 1 is bucket b's last tile.
template <typename KT, int DIG, typename C = Pass16Cfg<2>>
__global__ __launch_bounds__(C::BLOCK, C::WGS * 4) void rsx_pass16_kernel(const KT *__restrict__ kin, const KT *__restrict__ kin_hi,
                                                                         u32 lo_slots, unsigned short *__restrict__ kout,
                                                                         const SegTile *__restrict__ tiles,
                                                                         const u32 *__restrict__ btile,
                                                                         const SegCtl *__restrict__ ctl,
                                                                         const Plan *__restrict__ plan, u32 *__restrict__ cursors,
                                                                         u32 slack_cap, u32 *__restrict__ overflow, KdfArgs<KT> ka,
                                                                         u32 dbg = 0)
{
	static_assert(sizeof(KT) == 4, "4-byte keys, two-byte values");
	constexpr int BLOCK = C::BLOCK, KPT = C::KPT, TILE = C::TILE;
	if (ctl->blind != BLIND_GO || plan->hyb != HYB_TWO_LEVEL)
		return;   // (the attempt has been called off: rsx_hybrid.hpp)
	const u32 t = blockIdx.x;
	if (t >= ctl->ntiles)
		return;
	const SegTile st = tiles[t];
	const u32 cnt = st.cnt, bucket = st.bucket;
	const u32 shift = ctl->shift2;
	const u32 last_tile = btile[bucket + 1] - 1u;
	const KT *p = ((kin_hi && bucket >= lo_slots) ? kin_hi : kin) + st.beg;
	__shared__ Pass16Smem<C::WGS> sm;
	const u32 tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
	if (tid < 256)
		sm.cell[tid] = 0;
	__syncthreads();

	// ---- the tile's keys (any key in any lane) and the digits' counts
	KT keep[KPT];
	const bool full = cnt == (u32)TILE;
	if (full && (((uintptr_t)p) & 15) == 0) {
		typedef KT vec_t __attribute__((ext_vector_type(4)));
		const vec_t *vp = (const vec_t *)p + tid;
#pragma unroll
		for (int i = 0; i < KPT / 4; ++i) {
			const vec_t v = vp[i * BLOCK];
#pragma unroll
			for (int e = 0; e < 4; ++e)
				keep[4 * i + e] = v[e];
		}
	} else {
#pragma unroll
		for (int r = 0; r < KPT; ++r) {
			const u32 o = tid + r * BLOCK;
			keep[r] = o < cnt ? p[o] : (KT)0;
		}
	}
	if constexpr (DIG != 1) {   // (DIG_PLAIN: the key is its own derived key)
#pragma unroll
		for (int r = 0; r < KPT; ++r)
			keep[r] = kdf_apply(keep[r], ka);   // once: digits and values are bit fields of the derived key from here on
	}
	// (which of a lane's registers hold keys that exist: whole tiles all; a partial tile read by elements: o = tid + r * BLOCK.
	// The two cases are two copies of the loops: in whole tiles nothing is predicated)
	auto count = [&](auto full_c) {
		constexpr bool FULL = decltype(full_c)::value;
#pragma unroll
		for (int r = 0; r < KPT; ++r) {
			if (FULL || tid + r * BLOCK < cnt) {
				const u32 d = (u32)(keep[r] >> shift) & 0xFFu;
				atomicAdd(&sm.cell[d], 1u);
			}
		}
	};
	if (full)
		count(std::true_type{});
	else
		count(std::false_type{});
	__syncthreads();

	// ---- digit thread d: the run's place in the tile (scan) and in its slot (one returning atomic on the slot's cursor); the table
	// of the groups' digits
	u32 c = 0, tb = 0, excl = 0;
	if (tid < 256) {
		c = sm.cell[tid];
		if (c)   // (issued first: it crosses the fabric while the scan and the table are made)
			excl = __hip_atomic_fetch_add(cursors + ((u64)last_tile * 256u + tid), c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		u32 x = c;
#pragma unroll
		for (int off = 1; off < 64; off <<= 1) {
			const u32 y = __shfl_up(x, off);
			if (lane >= (u32)off)
				x += y;
		}
		if (lane == 63)
			sm.wsum[wid] = x;
		tb = x - c;
	}
	__syncthreads();
	if (tid < 256) {
		for (u32 k = 0; k < wid; ++k)
			tb += sm.wsum[k];
		sm.cell[tid] = tb;
		sm.rend[tid] = (unsigned short)(tb + c);   // (at most 32768)
		for (u32 g = (tb + 7u) >> 3; g < (tb + c + 7u) >> 3; ++g)
			sm.group_digit[g] = (unsigned char)tid;
	}
	__syncthreads();

	// ---- stage: the returning atomic on the run's cursor is the value's place; the digit threads first put down where their runs go
	if (tid < 256) {
		u64 running = ((u64)bucket * 256u + tid) * slack_cap + excl;
		if (excl + c > slack_cap) {
			// the slot is too small: the attempt will be discarded (rsx_seg_slack_plan_kernel sees the flag); the run goes to the dump
			// area behind the last slot (a tile of padding)
			atomicOr(overflow, 1u);
			running = (u64)65536u * slack_cap;
		}
		sm.delta[tid] = (u32)running - tb;
	}
	auto stage = [&](auto full_c) {
		constexpr bool FULL = decltype(full_c)::value;
		// (the digits are computed again, with a copy of the shift the compiler cannot see through: kept from the count phase
		// across the barriers they are 32 more registers per lane, and two workgroups per CU leave a lane 64)
		u32 shift_b = shift;
		asm volatile("" : "+s"(shift_b));
#pragma unroll
		for (int r0 = 0; r0 < KPT; r0 += 8) {
			// (eight atomics in flight before the first place is needed)
			u32 pos[8];
#pragma unroll
			for (int r = 0; r < 8; ++r) {
				pos[r] = 0;
				if (FULL || tid + (r0 + r) * BLOCK < cnt) {
					const u32 d = (u32)(keep[r0 + r] >> shift_b) & 0xFFu;
					pos[r] = __hip_atomic_fetch_add(&sm.cell[d], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
				}
			}
#pragma unroll
			for (int r = 0; r < 8; ++r) {
				if (FULL || tid + (r0 + r) * BLOCK < cnt) {
					*(unsigned short *)((char *)sm.stage + stage_swz<true>(pos[r] * 2u)) = (unsigned short)keep[r0 + r];
				}
			}
		}
	};
	if (full)
		stage(std::true_type{});
	else
		stage(std::false_type{});
	__syncthreads();

	// ---- write out: eight consecutive staged values per lane and step
#pragma unroll 1
	for (u32 i0 = 8u * tid; i0 < cnt; i0 += 8u * BLOCK) {
		const u32x4 x = *(const u32x4 *)((const char *)sm.stage + stage_swz<true>(i0 * 2u));
		u32 d = sm.group_digit[i0 >> 3];
		const u32 re = sm.rend[d];
		if (dbg) {
			// probe only (RSX_PASS16_DBG, wrong output): 1 = no global stores at all; 2 = what a pass whose runs are whole 64-byte atoms
			// would store -- the groups of the 64-byte blocks that lie inside ONE run, at 64-byte-aligned places of the run's slot
			if (dbg == 2) {
				const u32 b0 = i0 & ~31u;
				if (sm.group_digit[b0 >> 3] == d && b0 + 32u <= re && b0 + 32u <= cnt) {
					typedef u32x4 avec_t __attribute__((aligned(16)));
					*(avec_t *)(kout + (u32)(((sm.delta[d] + b0) & ~31u) + (i0 - b0))) = x;
				}
			}
			continue;
		}
		if (i0 + 8u <= re && i0 + 8u <= cnt) {
			typedef u32x4 uvec_t __attribute__((aligned(2)));
			*(uvec_t *)(kout + (u32)(sm.delta[d] + i0)) = x;
		} else {
			// a group across a run boundary (one in sixteen), or the tile's last: value by value
			const u64 lo = ((u64)x[1] << 32) | x[0], hi = ((u64)x[3] << 32) | x[2];
#pragma unroll 1
			for (u32 e = 0; e < 8u && i0 + e < cnt; ++e) {
				const u32 pos = i0 + e;
				while ((u32)sm.rend[d] <= pos)   // (empty runs end where they begin: walked over)
					++d;
				kout[(u32)(sm.delta[d] + pos)] = (unsigned short)((e < 4u ? lo : hi) >> (16u * (e & 3u)));
			}
		}
	}
}


// ---- the same pass with WHOLE 64-BYTE ATOMS ---------------------------------------------------------------------------------
// What the kernel above leaves on the table (RSX_PASS16_DBG, profiles/r05/pass16_store_probe.txt): 0.433 ms as it is, 0.245
// without its global stores, 0.320 when only whole, aligned 64-byte atoms are stored (76 % of the values) -- the RAGGED ends
// of the runs are what costs: a run of digit d of one tile ends inside a 64-byte atom whose other part a different CU
// writes at another time, and the memory side pays for a partly written atom about as much as for four whole ones
// (tools/ubench/store_spacing.hip: 512-byte runs at 3.2 TB/s shifted by 4 bytes, 6.0 TB/s aligned; where the 256 destinations
// lie -- 4 MiB apart, 64 MiB apart, inside one 2 MiB window -- moves that by a tenth: it is not translation).  Every pass
// kernel of rounds 1-4 wrote ragged runs, because a run's place followed from the keys of the tiles before it.
//
// Here a workgroup owns a contiguous RANGE of tiles of one level-1 bucket and CARRIES what does not fill an atom: per digit up
// to 31 values wait in the LDS for the next tile.  A tile's values of digit d first complete the carried atom, then go out
// as whole atoms, and the last < 32 are carried on.  A digit's place in its slot is a multiple of 32 values, taken from the
// slot's cursor by one returning global atomic per tile and digit as above -- so EVERY store of the pass is a whole, aligned
// 64-byte atom.  What is still carried when the range ends (or the bucket changes) goes to the last 128 values of the slot
// (its own cursor): the leaves read a slot's front and its back (rsx_leaf16_kernel; LeafSeg::ncols >> 16 = the back's count).
// The staged runs are laid out so that the part that goes out as atoms starts on a 16-byte boundary of the LDS: every lane of
// the write-out moves one aligned quarter atom, there is no value-by-value path.
// granule == 1 (SegCtl::leaf16 == 0: the leaves will be rsx_leaf_sort_kernel's, which read dense slots): nothing is carried,
// runs go out as they are -- the kernel above, inside this one's loop.
struct Pass16aCfg {
	static constexpr int BLOCK = 1024, KPT = 24, TILE = BLOCK * KPT;
	static constexpr u32 ATOM = 32;    // values per 64-byte atom
	static constexpr u32 BACK = 128;   // values at the end of every slot for what is carried when a range ends (LEAF16_BACK, rsx_leaf16.hpp):
	                                   // up to four workgroups' 31 per digit -- a bucket's tiles are shared by three at most
	static constexpr int STAGE = TILE + 256 * 14;   // + what the 16-byte alignment of 256 runs can cost
};

struct Pass16aSmem {
	__attribute__((aligned(16))) unsigned short stage[Pass16aCfg::STAGE];
	__attribute__((aligned(16))) unsigned short carry[256][32];
	u32 cell[2][256];   // per digit: count, then the run's cursor (tile-local); tiles alternate between the two (the other is zeroed meanwhile)
	u32 delta[256];     // slot position of a body value minus its tile-local position
	u32 info[256];      // carried before (6 bits) | head (6) | tail (6) | atom completed (1), for the copying threads
	unsigned short rbeg[256], bbeg[256], bend[256];   // the run, and the part of it that goes out as whole atoms (the body)
	unsigned char group_digit[Pass16aCfg::STAGE / 8];
	u32 wsum[4];
};

// cursors: [65536] front cursors, then [65536] back cursors (zeroed by rsx_blind_precheck_kernel with the status words).
template <typename KT, int DIG, bool PREFETCH = false>
__global__ __launch_bounds__(Pass16aCfg::BLOCK, 8) void rsx_pass16a_kernel(const KT *__restrict__ kin, const KT *__restrict__ kin_hi,
                                                                          u32 lo_slots, unsigned short *__restrict__ kout,
                                                                          const SegTile *__restrict__ tiles,
                                                                          const SegCtl *__restrict__ ctl,
                                                                          const Plan *__restrict__ plan, u32 *__restrict__ cursors,
                                                                          u32 slack_cap, u32 *__restrict__ overflow, KdfArgs<KT> ka)
{
	static_assert(sizeof(KT) == 4, "4-byte keys, two-byte values");
	typedef Pass16aCfg C;
	constexpr int BLOCK = C::BLOCK, KPT = C::KPT, TILE = C::TILE;
	if (ctl->blind != BLIND_GO || plan->hyb != HYB_TWO_LEVEL)
		return;
	const u32 ntiles = ctl->ntiles;
	const u32 per = (ntiles + gridDim.x - 1) / gridDim.x;
	const u32 t0 = blockIdx.x * per, t1 = t0 + per < ntiles ? t0 + per : ntiles;
	if (t0 >= t1)
		return;
	const u32 shift = ctl->shift2;
	const u32 gran = ctl->leaf16 ? C::ATOM : 1u;
	__shared__ Pass16aSmem sm;
	const u32 tid0 = threadIdx.x;
	auto sidx = [](u32 pos) { return stage_swz<true>(pos * 2u); };   // byte offset of staged value `pos`
	auto staged = [&](u32 pos) -> unsigned short & { return *(unsigned short *)((char *)sm.stage + sidx(pos)); };
	u32 cc = 0;                // digit thread: values of its digit carried from the tiles before
	u32 bucket = tiles[t0].bucket;
	if (tid0 < 256)
		sm.cell[0][tid0] = 0;
	// what is carried goes to the back of its slot (the range ends, or the next tile lies in another bucket)
	auto flush = [&]() {
		const u32 tid = tid0, cd = tid >> 2, part = tid & 3u;
		__syncthreads();   // (the tile before is through: its tables are free, what it carries on is in place)
		if (tid < 256) {
			u32 inf = 0, dest = 0;
			if (cc) {
				const u32 slot = bucket * 256u + tid;
				const u32 pos = __hip_atomic_fetch_add(cursors + 65536u + slot, cc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				if (pos + cc > C::BACK)
					atomicOr(overflow, 1u);
				else {
					inf = cc;
					dest = slot * slack_cap + (slack_cap - C::BACK) + pos;
				}
			}
			sm.info[tid] = inf;
			sm.delta[tid] = dest;
			cc = 0;
		}
		__syncthreads();
		{
			const u32 n = sm.info[cd], dest = sm.delta[cd];
#pragma unroll 2
			for (u32 e = 0; e < 8; ++e) {
				const u32 k = part * 8u + e;
				if (k < n)
					kout[dest + k] = sm.carry[cd][k];
			}
		}
		__syncthreads();
	};
	__syncthreads();
	KT keep[KPT];
	// the keys of tile t into keep[] (as they lie in memory; only issued here)
	auto request = [&](const u32 t, const u32 tid) {
		const SegTile st = tiles[t];
		const KT *p = ((kin_hi && st.bucket >= lo_slots) ? kin_hi : kin) + st.beg;
		if (st.cnt == (u32)TILE && (((uintptr_t)p) & 15) == 0) {
			typedef KT vec_t __attribute__((ext_vector_type(4)));
			const vec_t *vp = (const vec_t *)p + tid;
#pragma unroll
			for (int i = 0; i < KPT / 4; ++i) {
				// (non-temporal: the level-1 slots are read once; loads that do not displace the slot values this pass writes -- which the
				// leaves read next -- make the pass 3.5 % shorter: 0.396 -> 0.382 ms for 2^28 keys, five A/B rounds on one box,
				// profiles/r06/nontemporal_ab.txt.  The same hint on the level-1 pass's loads or stores costs 8-10 % there.)
				const vec_t v = __builtin_nontemporal_load(&vp[i * BLOCK]);
#pragma unroll
				for (int e = 0; e < 4; ++e)
					keep[4 * i + e] = v[e];
			}
		} else {
#pragma unroll
			for (int r = 0; r < KPT; ++r) {
				const u32 o = tid + r * BLOCK;
				keep[r] = o < st.cnt ? p[o] : (KT)0;
			}
		}
	};
	// (the two workgroups of a CU start half a tile apart: started together they would load, rank and store in step)
	if (blockIdx.x >= gridDim.x / 2)
		__builtin_amdgcn_s_sleep(127);   // (about 8 k cycles)
	// PREFETCH (probe): the next tile's keys are requested before this tile is written out.  With 64 registers per lane the
	// compiler then spills the arriving keys to scratch (54-85 dwords) -- the second workgroup of the CU is the overlap.
	if constexpr (PREFETCH)
		request(t0, tid0);
	for (u32 t = t0; t < t1; ++t) {
		// (everything a tile derives from the thread index is derived from an opaque copy of it, made per tile: as loop invariants
		// the LDS addresses of a dozen tables would be hoisted in front of the loop and spilled there)
		u32 tid = tid0;
		asm volatile("" : "+v"(tid));
		const u32 lane = tid & 63, wid = tid >> 6;
		const u32 cd = tid >> 2, part = tid & 3u;   // the copying threads: digit, quarter of an atom
		u32 *const cell = sm.cell[(t - t0) & 1u];
		const SegTile st = tiles[t];
		if (st.bucket != bucket) {
			flush();
			bucket = st.bucket;
		}
		const u32 cnt = st.cnt;
		const bool full = cnt == (u32)TILE;
		// ---- the tile's keys (any key in any lane), derived once, and the digits' counts
		if constexpr (!PREFETCH)
			request(t, tid);
		if constexpr (DIG != 1) {
#pragma unroll
			for (int r = 0; r < KPT; ++r)
				keep[r] = kdf_apply(keep[r], ka);
		}
		auto count = [&](auto full_c) {
			constexpr bool FULL = decltype(full_c)::value;
#pragma unroll
			for (int r = 0; r < KPT; ++r) {
				if (FULL || tid + r * BLOCK < cnt)
					atomicAdd(&cell[(u32)(keep[r] >> shift) & 0xFFu], 1u);
			}
		};
		if (full)
			count(std::true_type{});
		else
			count(std::false_type{});
		__syncthreads();

		// ---- digit thread d: what of (carried + this tile's) values goes out, where in the slot, where in the staging area
		// (only `base`, the answer of the global atomic, lives in a register across the next barriers: the rest is re-read from
		// the tables -- the persistent loop leaves a lane 64 registers, 24 of them keys)
		u32 base = 0;
		{
			u32 rlen = 0, rstart = 0;
			if (tid < 256) {
				const u32 c = cell[tid];
				u32 h, body = 0, tail = 0, atom = 0;
				const bool enough = cc + c >= gran;
				if (enough) {
					h = cc ? gran - cc : 0u;      // the head completes the carried atom
					atom = cc ? 1u : 0u;
					body = (c - h) & ~(gran - 1u);
					tail = (c - h) & (gran - 1u);
				} else {
					h = c;                        // too few for an atom: all of it joins the carried values
				}
				const u32 m = atom * gran + body;
				if (m)   // (issued first: it crosses the fabric while the layout is made)
					base = __hip_atomic_fetch_add(cursors + (bucket * 256u + tid), m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				const u32 o = (8u - (h & 7u)) & 7u;   // the run starts `o` values into its region: the body then starts on a 16-byte boundary
				rlen = (o + c + 7u) & ~7u;
				sm.info[tid] = cc | (h << 6) | (tail << 12) | (atom << 18) | ((enough ? 1u : 0u) << 19) | (o << 20);
				sm.bend[tid] = (unsigned short)body;   // (for now: the body's length)
				cc = enough ? tail : cc + c;
				u32 x = rlen;
#pragma unroll
				for (int off = 1; off < 64; off <<= 1) {
					const u32 y = __shfl_up(x, off);
					if (lane >= (u32)off)
						x += y;
				}
				if (lane == 63)
					sm.wsum[wid] = x;
				rstart = x - rlen;
			}
			__syncthreads();
			if (tid < 256) {
				for (u32 k = 0; k < wid; ++k)
					rstart += sm.wsum[k];
				const u32 inf = sm.info[tid];
				const u32 rb = rstart + (inf >> 20), bb = rb + ((inf >> 6) & 63u), be = bb + sm.bend[tid];
				cell[tid] = rb;
				sm.cell[((t - t0) & 1u) ^ 1u][tid] = 0;   // (the next tile's counters: last used as the cursors of the tile before, three barriers ago)
				sm.rbeg[tid] = (unsigned short)rb;
				sm.bbeg[tid] = (unsigned short)bb;
				sm.bend[tid] = (unsigned short)be;
				for (u32 g = bb >> 3; g < (be + 7u) >> 3; ++g)
					sm.group_digit[g] = (unsigned char)tid;
			}
		}
		__syncthreads();

		// ---- stage; the digit threads first put down where their bodies go
		if (tid < 256) {
			const u32 slot = bucket * 256u + tid;
			const u32 atom = (sm.info[tid] >> 18) & 1u, bb = sm.bbeg[tid], m = atom * gran + (sm.bend[tid] - bb);
			u32 dest = slot * slack_cap + base + atom * gran;   // of the body's first value
			if (m && base + m > slack_cap - C::BACK) {
				// the slot is too small: the attempt will be discarded (rsx_seg_slack_plan_kernel sees the flag); the values go to the
				// dump area behind the last slot (a tile of padding)
				atomicOr(overflow, 1u);
				dest = 65536u * slack_cap + C::ATOM;
			}
			sm.delta[tid] = dest - bb;
		}
		u32 shift_b = shift;
		asm volatile("" : "+s"(shift_b));   // (the digits are computed again, not kept across the barriers: 64 registers per lane)
		auto stage_keys = [&](auto full_c) {
			constexpr bool FULL = decltype(full_c)::value;
#pragma unroll
			for (int r0 = 0; r0 < KPT; r0 += 8) {
				u32 pos[8];
#pragma unroll
				for (int r = 0; r < 8; ++r) {
					pos[r] = 0;
					if (FULL || tid + (r0 + r) * BLOCK < cnt)
						pos[r] = __hip_atomic_fetch_add(&cell[(u32)(keep[r0 + r] >> shift_b) & 0xFFu], 1u, __ATOMIC_RELAXED,
						                                __HIP_MEMORY_SCOPE_WORKGROUP);
				}
#pragma unroll
				for (int r = 0; r < 8; ++r) {
					if (FULL || tid + (r0 + r) * BLOCK < cnt)
						staged(pos[r]) = (unsigned short)keep[r0 + r];
				}
			}
		};
		if (full)
			stage_keys(std::true_type{});
		else
			stage_keys(std::false_type{});
		__syncthreads();

		// the next tile's keys are requested now -- the registers are free, and they cross the memory system while this tile is
		// written out
		if constexpr (PREFETCH) {
			if (t + 1 < t1)
				request(t + 1, tid);
		}
		// ---- out: the completed atoms (a quarter per copying thread: carried values, then the head of the run) ...
		{
			const u32 inf = sm.info[cd];
			const u32 ccd = inf & 63u, hd = (inf >> 6) & 63u, atomd = (inf >> 18) & 1u;
			if (atomd) {
				const u32 rb = sm.rbeg[cd];
				u64 qlo = 0, qhi = 0;
				// (four values at a time: unrolled eight-fold, the addresses and values in flight cost the registers the next tile's
				// keys are arriving in)
#pragma unroll 1
				for (u32 half = 0; half < 2u; ++half) {
					u64 acc = 0;
#pragma unroll
					for (u32 e = 0; e < 4; ++e) {
						const u32 k = part * 8u + half * 4u + e;
						const u32 v = k < ccd ? (u32)sm.carry[cd][k] : (u32)staged(rb + (k - ccd));
						acc |= (u64)v << (16u * e);
					}
					if (half)
						qhi = acc;
					else
						qlo = acc;
				}
				(void)hd;
				typedef u32x4 avec_t __attribute__((aligned(16)));
				*(avec_t *)(kout + (u32)(sm.delta[cd] + sm.bbeg[cd] - C::ATOM + part * 8u)) =
				    u32x4{(u32)qlo, (u32)(qlo >> 32), (u32)qhi, (u32)(qhi >> 32)};
			}
		}
		// ... and the bodies: every group of eight staged values that lies in one is a quarter of an aligned atom
		// (granule 1: runs of any length -- the last group of a body goes value by value)
		{
			const u32 total = (u32)__builtin_amdgcn_readfirstlane((int)sm.wsum[0]) + sm.wsum[1] + sm.wsum[2] + sm.wsum[3];
#pragma unroll 1
			for (u32 i0 = 8u * tid; i0 < total; i0 += 8u * BLOCK) {
				const u32 d = sm.group_digit[i0 >> 3];
				const u32 bb = sm.bbeg[d], be = sm.bend[d];
				if (i0 >= bb && i0 < be) {
					const u32x4 x = *(const u32x4 *)((const char *)sm.stage + sidx(i0));
					if (i0 + 8u <= be) {
						typedef u32x4 uvec_t __attribute__((aligned(2)));
						*(uvec_t *)(kout + (u32)(sm.delta[d] + i0)) = x;
					} else {
						const u64 lo = ((u64)x[1] << 32) | x[0], hi = ((u64)x[3] << 32) | x[2];
#pragma unroll 1
						for (u32 e = 0; i0 + e < be; ++e)
							kout[(u32)(sm.delta[d] + i0 + e)] = (unsigned short)((e < 4u ? lo : hi) >> (16u * (e & 3u)));
					}
				}
			}
		}
		// ---- what stays: the tail of the run (or, with too few values for an atom, all of the run behind what was carried).  (No
		// barrier in front: carry[d][8 part ..] was read for the atom above by this very thread.  None behind: the next tile stages
		// -- and reads the carried values -- behind three barriers of its own.)
		{
			const u32 inf = sm.info[cd];
			const u32 ccd = inf & 63u, hd = (inf >> 6) & 63u, taild = (inf >> 12) & 63u, fulld = (inf >> 19) & 1u;
			const u32 from = fulld ? sm.bend[cd] : sm.rbeg[cd], to = fulld ? 0u : ccd, n = fulld ? taild : hd;
#pragma unroll 2
			for (u32 e = 0; e < 8; ++e) {
				const u32 k = part * 8u + e;
				if (k < n)
					sm.carry[cd][to + k] = staged(from + k);
			}
		}
	}
	flush();
}

}  // namespace rsx
