// rsx_pass32.hpp -- the LEVEL-1 pass of a keys-only sort of 4-byte keys without a histogram, in whole 64-byte atoms (round 5).
//
// The pass (rsx_hybrid.hpp): every key of the caller's array goes to slot d of 256, d = its level-1 digit
// (SegCtl::shift1: the highest kept column, radix_sort.hpp:82-90's last trip made first, README.md:647-650), as its untouched
// element image; the keys that differ from the first one in a column the sample took for constant call the attempt off
// (SegCtl::cmask, radix_sort.hpp:64-70).  Through round 4 this was rsx_scatter2_kernel<..., SEG> with SCATTER_BLIND_TOP.
//
// What rsx_pass16a_kernel (rsx_pass16.hpp) does for the level-2 pass, for 4-byte values: a workgroup owns a contiguous range
// of tiles and CARRIES, per digit, the up to 15 keys that do not fill a 64-byte atom; a tile's keys of a digit complete the
// carried atom, go out as whole atoms, and the rest is carried on; a digit's place in its slot is a multiple of 16 keys,
// taken from the slot's cursor by one returning global atomic per tile and digit.  No look-back chain (the order of the tiles
// inside a slot is free: the leaves sort), no partly written atom anywhere -- the ragged run ends of the chained pass were what
// its stores cost (tools/ubench/store_spacing.hip; the level-2 pass: 0.458 -> 0.397 ms for 2^28 keys).  What a workgroup still
// carries when its range ends goes to the last PASS32_BACK keys of the slot (a cursor of its own): a bucket then lies at both
// ends of its slot and rsx_seg_tiles_kernel cuts the level-2 pass's tiles from both.
// One workgroup per CU (150 KB of LDS: a 28 Ki-key tile, the carried keys, the tables); the next tile's keys are requested
// while this one is written out.
#pragma once

#include "rsx_kernels.hpp"
#include "rsx_hybrid.hpp"
#include "rsx_scatter2.hpp"

namespace rsx {

template <int KPT_ = 28, bool RANK1_ = true> struct Pass32aCfgT {
	static constexpr int BLOCK = 1024, KPT = KPT_, TILE = BLOCK * KPT;
	// a key's place inside its digit's run is the value the COUNTING atomic returned (kept, sixteen bits per key): the staging phase
	// reads the run's start and adds it -- one LDS atomic per key and tile instead of two (false: count, then a second returning atomic
	// on the run's cursor, as rsx_pass16a_kernel does)
	static constexpr bool RANK1 = RANK1_;
	static constexpr int WPE = KPT_ <= 12 ? 8 : 4;   // (12 keys per lane: 81 KB of LDS, two workgroups per CU, 64 registers)
	static constexpr int SB = KPT_ % 7 == 0 ? 7 : 6;  // staging atomics in flight
	static constexpr u32 ATOM = 16;     // 4-byte keys per 64-byte atom (8-byte keys: 8 -- the kernel's own ATOM)
	static constexpr u32 BACK = 8192;   // keys at the end of every slot for what is carried when a range ends (up to 512 workgroups x 15)
	static constexpr int STAGE = TILE + 256 * 6;   // + what the 16-byte alignment of 256 runs can cost
};
typedef Pass32aCfgT<28> Pass32aCfg;
constexpr u32 PASS32_BACK = Pass32aCfg::BACK;

template <typename KT, typename C, int REP = 1> struct Pass32aSmem {
	__attribute__((aligned(16))) KT stage[C::STAGE];
	__attribute__((aligned(16))) KT carry[256][64 / sizeof(KT)];
	__attribute__((aligned(16))) u32 cell[2][256 * REP];   // per (digit, lane class): count, then the class's part of the run (tile-local start / cursor); tiles alternate between the two
	u32 delta[256];     // slot position of a body key minus its tile-local position
	u32 info[256];      // carried before (5 bits) | head (5) | tail (5) | atom completed | enough for an atom | offset in the region
	unsigned short rbeg[256], bbeg[256], bend[256];
	unsigned char group_digit[C::STAGE / (16 / sizeof(KT))];   // per 16-byte group of staged keys
	u32 wsum[4];
};

// kout: the lower of the two arrays the slots lie in; slot d starts (d < lo_slots ? off_lo : off_hi) + d * cap keys from there
// (SegArgs, rsx_scatter2.hpp).  cursors: [256] front cursors, [256] back cursors (zeroed by rsx_blind_precheck_kernel).
// REP: counters per digit, one per lane class (lane & (REP - 1)).  An LDS atomic serialises the lanes of one instruction that hit one
// word; keys that arrive in ORDER of their level-1 digit -- what the local sorts of a distributed sort receive from the senders'
// split passes, piece by piece (rsx_sort_inplace_async_hint) -- put all 64 lanes on one counter: 2^29 such keys took 4.5 ms on this
// route, no better than one pass per column.  With four counters per digit: 3.25 ms (one piece), 2.78 (two pieces: 3.42), 2.59
// (four: 2.88), 2.54 (eight: 2.62).  Evenly spread digits pay 1-3 % of this pass for the wider tables (0.430 -> 0.437-0.450 ms at
// 2^28 keys), so the hinted sorts take REP = 4 and everybody else REP = 1 (profiles/r06/presplit_probe.txt).
// OT (round 6): what a slot holds.  KT: the element images.  u32 with 8-byte keys: the low word of every DERIVED key -- where nothing
// below the level-1 digit varies above bit 32 (SegCtl::narrow == 2: keys below 2^40, say, BASELINE.json's cfg 3 (ii) / (iii)) the
// passes behind this one never look at more, and the leaves put the upper word back from the first key and the slot
// (rsx_leafk_kernel, SLOT32): 12 instead of 16 bytes per key through this pass, 8 instead of 12 through the next.  Both forms are
// enqueued; the sample decides which one works.
template <typename KT, int DIG, bool PREFETCH = true, typename C = Pass32aCfg, int REP_ = 1, typename OT = KT>
__global__ __launch_bounds__(C::BLOCK, C::WPE) void rsx_pass32a_kernel(const KT *__restrict__ kin, u64 n, OT *__restrict__ kout,
                                                                          u32 lo_slots, u32 off_lo, u32 off_hi, u32 cap,
                                                                          const SegCtl *__restrict__ ctl,
                                                                          u32 *__restrict__ cursors, u32 *__restrict__ overflow,
                                                                          KdfArgs<KT> ka)
{
	static_assert(sizeof(KT) == 4 || sizeof(KT) == 8, "4- or 8-byte keys");
	constexpr int BLOCK = C::BLOCK, KPT = C::KPT, TILE = C::TILE, SB = C::SB;
	constexpr u32 VIN = 16 / sizeof(KT);    // keys per 16-byte load
	constexpr u32 VEC = 16 / sizeof(OT);    // values per 16-byte vector (what a copying thread moves: a quarter of an atom)
	constexpr u32 ATOM = 64 / sizeof(OT);   // values per 64-byte atom
	constexpr bool NARROW = sizeof(OT) != sizeof(KT);
	static_assert(!NARROW || (sizeof(KT) == 8 && sizeof(OT) == 4), "8-byte keys into four-byte slots, or the element images");
	static_assert(KPT % (int)VIN == 0, "whole vectors per lane");
	static_assert(sizeof(OT) == 4 || C::STAGE * sizeof(OT) + 256 * 64 <= 150 * 1024, "8-byte slots: 14 Ki-key tiles (Pass32aCfgT<14>)");
	if (ctl->blind != BLIND_GO)
		return;   // (the sample has called the attempt off: rsx_hybrid.hpp)
	if constexpr (sizeof(KT) == 8) {
		if ((ctl->narrow == 2u) != NARROW)
			return;   // (the other form's keys)
	}
	const u32 ntiles = (u32)((n + TILE - 1) / TILE);
	const u32 per = (ntiles + gridDim.x - 1) / gridDim.x;
	const u32 t0 = blockIdx.x * per, t1 = t0 + per < ntiles ? t0 + per : ntiles;
	if (t0 >= t1)
		return;
	const u32 shift = ctl->shift1;
	// the bits the sample took for constant, and the first key's (derived)
	const KT cmask = sizeof(KT) == 8 ? (KT)(((u64)ctl->cmask_hi << 32) | ctl->cmask_lo) : (KT)ctl->cmask_lo;
	const KT key0 = sizeof(KT) == 8 ? (KT)(((u64)ctl->key0_hi << 32) | ctl->key0_lo) : (KT)ctl->key0_lo;
	__shared__ Pass32aSmem<OT, C, REP_> sm;
	const u32 tid0 = threadIdx.x;
	auto sidx = [](u32 pos) { return stage_swz<true>(pos * (u32)sizeof(OT)); };
	auto staged = [&](u32 pos) -> OT & { return *(OT *)((char *)sm.stage + sidx(pos)); };
	auto slot_base = [&](u32 d) { return (d < lo_slots ? off_lo : off_hi) + d * cap; };
	u32 cc = 0;   // digit thread: keys of its digit carried from the tiles before
	constexpr u32 REP = REP_;
	static_assert(REP == 1 || REP == 4, "one counter per digit, or one per lane class");
	if (tid0 < 256 * REP)
		sm.cell[0][tid0] = 0;
	__syncthreads();
	auto bucket_of = [&](KT k, u32 sh) -> u32 { return (u32)(k >> sh) & 0xFFu; };
	KT keep[KPT];
	auto request = [&](const u32 t, const u32 tid) {
		const u64 beg = (u64)t * TILE;
		const u32 cnt = n - beg < (u64)TILE ? (u32)(n - beg) : (u32)TILE;
		const KT *p = kin + beg;
		if (cnt == (u32)TILE && (((uintptr_t)p) & 15) == 0) {
			typedef KT vec_t __attribute__((ext_vector_type(VIN)));
			const vec_t *vp = (const vec_t *)p + tid;
#pragma unroll
			for (int i = 0; i < KPT / (int)VIN; ++i) {
				const vec_t v = vp[i * BLOCK];
#pragma unroll
				for (int e = 0; e < (int)VIN; ++e)
					keep[(int)VIN * i + e] = v[e];
			}
		} else {
#pragma unroll
			for (int r = 0; r < KPT; ++r) {
				const u32 o = tid + r * BLOCK;
				keep[r] = o < cnt ? p[o] : (KT)0;
			}
		}
	};
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
		const u64 beg = (u64)t * TILE;
		const u32 cnt = n - beg < (u64)TILE ? (u32)(n - beg) : (u32)TILE;
		const bool full = cnt == (u32)TILE;
		if constexpr (!PREFETCH)
			request(t, tid);
		// ---- the digits' counts; every key against the columns the sample took for constant
		u32 rk[C::RANK1 ? (KPT + 1) / 2 : 1];   // RANK1: the keys' ranks in their digits' runs, two to a register
		auto count = [&](auto full_c) {
			constexpr bool FULL = decltype(full_c)::value;
			KT bad = 0;
#pragma unroll
			for (int r = 0; r < KPT; ++r) {
				u32 mine = 0;
				if (FULL || tid + r * BLOCK < cnt) {
					const KT k = DIG == 1 ? keep[r] : kdf_apply(keep[r], ka);
					bad |= (k ^ key0) & cmask;
					const u32 b = bucket_of(k, shift) * REP + (lane & (REP - 1u));
					if constexpr (C::RANK1)
						mine = __hip_atomic_fetch_add(&cell[b], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
					else
						atomicAdd(&cell[b], 1u);
				}
				if constexpr (C::RANK1)
					rk[r >> 1] = (r & 1) ? rk[r >> 1] | (mine << 16) : mine;
			}
			if (__ballot(bad != 0) && lane == 0)
				atomicOr(overflow, 1u);   // (a column that is not constant after all: the attempt is lost)
		};
		if (full)
			count(std::true_type{});
		else
			count(std::false_type{});
		__syncthreads();

		// ---- digit thread d: what of (carried + this tile's) keys goes out, where in the slot, where in the staging area
		u32 base = 0;
		{
			u32 rlen = 0, rstart = 0;
			u32 crep[REP];   // digit thread: its digit's keys per lane class
			if (tid < 256) {
				u32 c = 0;
#pragma unroll
				for (u32 q = 0; q < REP; ++q) {
					crep[q] = cell[tid * REP + q];
					c += crep[q];
				}
				u32 h, body = 0, tail = 0, atom = 0;
				const bool enough = cc + c >= ATOM;
				if (enough) {
					h = cc ? ATOM - cc : 0u;   // the head completes the carried atom
					atom = cc ? 1u : 0u;
					body = (c - h) & ~(ATOM - 1u);
					tail = (c - h) & (ATOM - 1u);
				} else {
					h = c;                        // too few for an atom: all of it joins the carried keys
				}
				const u32 m = atom * ATOM + body;
				if (m)   // (issued first: it crosses the fabric while the layout is made)
					base = __hip_atomic_fetch_add(cursors + tid, m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				const u32 o = (VEC - (h & (VEC - 1u))) & (VEC - 1u);   // the run starts `o` keys into its region: the body then starts on a 16-byte boundary
				rlen = (o + c + VEC - 1u) & ~(VEC - 1u);
				sm.info[tid] = cc | (h << 5) | (tail << 10) | (atom << 15) | ((enough ? 1u : 0u) << 16) | (o << 17);
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
				const u32 rb = rstart + (inf >> 17), bb = rb + ((inf >> 5) & 31u), be = bb + sm.bend[tid];
				u32 rq = rb;
#pragma unroll
				for (u32 q = 0; q < REP; ++q) {
					cell[tid * REP + q] = rq;   // (the classes' parts of the run, one behind the other)
					rq += crep[q];
					sm.cell[((t - t0) & 1u) ^ 1u][tid * REP + q] = 0;   // (the next tile's counters: last used as the cursors of the tile before)
				}
				sm.rbeg[tid] = (unsigned short)rb;
				sm.bbeg[tid] = (unsigned short)bb;
				sm.bend[tid] = (unsigned short)be;
				for (u32 g = bb / VEC; g < (be + VEC - 1u) / VEC; ++g)
					sm.group_digit[g] = (unsigned char)tid;
			}
		}
		__syncthreads();

		// ---- stage; the digit threads first put down where their bodies go
		if (tid < 256) {
			const u32 atom = (sm.info[tid] >> 15) & 1u, bb = sm.bbeg[tid], m = atom * ATOM + (sm.bend[tid] - bb);
			u32 dest = slot_base(tid) + base + atom * ATOM;   // of the body's first key
			if (m && base + m > cap - C::BACK) {
				// the slot is too small: the attempt will be discarded (rsx_seg_tiles_kernel sees the flag).  Its keys go over the
				// slot's own beginning -- a slot holds more than a tile (blind_enqueue) and nothing of a lost attempt is read
				atomicOr(overflow, 1u);
				dest = slot_base(tid) + ATOM;
			}
			sm.delta[tid] = dest - bb;
		}
		u32 shift_b = shift;
		asm volatile("" : "+s"(shift_b));   // (the digits are computed again, not kept across the barriers)
		auto stage_keys = [&](auto full_c) {
			constexpr bool FULL = decltype(full_c)::value;
#pragma unroll
			for (int r0 = 0; r0 < KPT; r0 += SB) {
				u32 pos[SB];
#pragma unroll
				for (int r = 0; r < SB; ++r) {
					pos[r] = 0;
					if (FULL || tid + (r0 + r) * BLOCK < cnt) {
						const KT k = DIG == 1 ? keep[r0 + r] : kdf_apply(keep[r0 + r], ka);
						const u32 b = bucket_of(k, shift_b) * REP + (lane & (REP - 1u));
						if constexpr (C::RANK1)
							pos[r] = cell[b] + ((rk[(r0 + r) >> 1] >> (16 * ((r0 + r) & 1))) & 0xFFFFu);   // (the run's start: nobody moves it)
						else
							pos[r] = __hip_atomic_fetch_add(&cell[b], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
					}
				}
#pragma unroll
				for (int r = 0; r < SB; ++r) {
					if (FULL || tid + (r0 + r) * BLOCK < cnt) {
						if constexpr (NARROW)
							staged(pos[r]) = (OT)(DIG == 1 ? keep[r0 + r] : kdf_apply(keep[r0 + r], ka));
						else
							staged(pos[r]) = keep[r0 + r];
					}
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

		// ---- out: the completed atoms (a quarter per copying thread: carried keys, then the head of the run) ...
		{
			const u32 inf = sm.info[cd];
			const u32 ccd = inf & 31u, atomd = (inf >> 15) & 1u;
			if (atomd) {
				const u32 rb = sm.rbeg[cd];
				typedef OT kvec_t __attribute__((ext_vector_type(VEC)));
				typedef kvec_t avec_t __attribute__((aligned(16)));
				kvec_t w;
#pragma unroll
				for (u32 e = 0; e < VEC; ++e) {
					const u32 k = part * VEC + e;
					w[e] = k < ccd ? sm.carry[cd][k] : staged(rb + (k - ccd));
				}
				*(avec_t *)(kout + (u32)(sm.delta[cd] + sm.bbeg[cd] - ATOM + part * VEC)) = w;
			}
		}
		// ... and the bodies: every group of four staged keys that lies in one is a quarter of an aligned atom
		{
			const u32 total = (u32)__builtin_amdgcn_readfirstlane((int)sm.wsum[0]) + sm.wsum[1] + sm.wsum[2] + sm.wsum[3];
#pragma unroll 1
			for (u32 i0 = VEC * tid; i0 < total; i0 += VEC * BLOCK) {
				const u32 d = sm.group_digit[i0 / VEC];
				if (i0 >= sm.bbeg[d] && i0 < sm.bend[d]) {
					typedef OT kvec_t __attribute__((ext_vector_type(VEC)));
					typedef kvec_t avec_t __attribute__((aligned(16)));
					*(avec_t *)(kout + (u32)(sm.delta[d] + i0)) = *(const kvec_t *)((const char *)sm.stage + sidx(i0));
				}
			}
		}
		// ---- what stays: the tail of the run (or, with too few keys for an atom, all of the run behind what was carried).  (No
		// barrier in front: carry[d][4 part ..] was read for the atom above by this very thread.  None behind: the next tile stages
		// -- and reads the carried keys -- behind three barriers of its own.)
		{
			const u32 inf = sm.info[cd];
			const u32 ccd = inf & 31u, hd = (inf >> 5) & 31u, taild = (inf >> 10) & 31u, enoughd = (inf >> 16) & 1u;
			const u32 from = enoughd ? sm.bend[cd] : sm.rbeg[cd], to = enoughd ? 0u : ccd, nk = enoughd ? taild : hd;
#pragma unroll
			for (u32 e = 0; e < VEC; ++e) {
				const u32 k = part * VEC + e;
				if (k < nk)
					sm.carry[cd][to + k] = staged(from + k);
			}
		}
	}
	// ---- what is still carried goes to the back of its slot
	{
		const u32 tid = tid0, cd = tid >> 2, part = tid & 3u;
		__syncthreads();
		if (tid < 256) {
			u32 inf = 0, dest = 0;
			if (cc) {
				const u32 pos = __hip_atomic_fetch_add(cursors + 256u + tid, cc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				if (pos + cc > C::BACK)
					atomicOr(overflow, 1u);
				else {
					inf = cc;
					dest = slot_base(tid) + (cap - C::BACK) + pos;
				}
			}
			sm.info[tid] = inf;
			sm.delta[tid] = dest;
		}
		__syncthreads();
		const u32 nk = sm.info[cd], dest = sm.delta[cd];
#pragma unroll
		for (u32 e = 0; e < VEC; ++e) {
			const u32 k = part * VEC + e;
			if (k < nk)
				kout[dest + k] = sm.carry[cd][k];
		}
	}
}

}  // namespace rsx
