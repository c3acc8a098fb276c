// rsx_scatter2.hpp -- the "count first" scatter pass (default), gfx950.
//
// Same job as rsx_scatter_kernel (rsx_kernels.hpp): one stable scatter pass of the reference's
// loop at radix_sort.hpp:82-90.  What is different is the order of work inside a workgroup:
//
//   phase A  the workgroup takes a super-tile of up to TPS tiles and COUNTS first: one streaming
//            read of its keys (from HBM) fills, for every tile, a 16-bit cell per (wave, digit).
//            From the cells come the super-tile's digit totals (published, chained by decoupled
//            look-back exactly as in rsx_scatter_kernel) and, by prefix sums over waves and digits,
//            the tile-local start of every (wave, digit) run, which is written back into the cells.
//   phase B  per tile, a second read of the keys (now out of L2 / Infinity Cache): one returning
//            LDS atomic add on the (wave, digit) cell hands each key its final tile-local position
//            and the key is staged there at once -- no rank registers, no rank/stage barrier, and
//            the register footprint does not grow with the tile.  Then the staged tile is written
//            out in wide chunks.
//
// Why tiles are as large as LDS allows (32 Ki four-byte keys = 128 KiB of staging, one workgroup
// of 16 waves per CU): on MI355X the memory side retires one write request per touched 256-byte
// block, so a scattered write is only efficient when each digit's run is long.  With 256 digits a
// 32 Ki-key tile gives 512-byte runs (tools/ubench/store_runs.hip: 128-byte runs 2.2 TB/s, 512-byte
// runs 3.0 TB/s, 1 KiB 3.8 TB/s, unaligned).
//
// Stability rests on the returning LDS atomic: lanes of one wave-instruction that hit the same cell
// must receive their return values in increasing lane order, and successive instructions of a wave
// in issue order.  gfx950 does that (tools/ubench/lds_atomic_order.hip) but it is not a documented
// guarantee, so the host checks it on the device before it selects this kernel (rsx.hip).
#pragma once

#include "rsx_kernels.hpp"
#include "rsx_hybrid.hpp"

namespace rsx {

// KPT_ = 0: the default tile (as large as LDS allows).  Arrays of up to a few million keys take a quarter of it
// (Sc2SmallCfg): with default tiles most CUs would have nothing to do, and the output is cache-resident there anyway,
// so short runs cost nothing.
template <typename KT, typename VT, int NWAVES_ = 16, int TPS_ = 1, int LB_ = 8, bool CELL16_ = false, int KPT_ = 0> struct Sc2Cfg {
	static constexpr bool CELL16 = CELL16_;           // 16-bit cells packed two per word, or one 32-bit cell per digit
	static constexpr int NWAVES = NWAVES_;
	static constexpr int BLOCK = NWAVES * 64;
	static constexpr int ELEM = sizeof(KT) > (size_t)val_bytes<VT>::value ? sizeof(KT) : val_bytes<VT>::value;
	static constexpr int KPT = KPT_ ? KPT_ : (ELEM == 8 ? 16 : 32);   // keys per lane: 128 KiB of staging at 16 waves
	static constexpr int TILE = BLOCK * KPT;
	static constexpr int TPS = TPS_;                   // tiles per super-tile (at most)
	static constexpr int LB = LB_;                     // status words fetched per look-back round trip
	static constexpr int SB = KPT >= 16 ? 8 : KPT / 2;  // keys per lane in flight in the streaming loops
	static constexpr int CHUNK = 16 / ELEM;            // consecutive staged elements one lane writes out together
	static constexpr int STAGE_BYTES = TILE * ELEM;
	static_assert(NWAVES >= 4, "256 digit threads are needed");
	static_assert(TILE <= 32768, "tile-local positions live in 16-bit cells, two per word");
	static_assert(KPT % (2 * SB) == 0 && KPT % CHUNK == 0 && KPT % 2 == 0, "whole batch pairs / chunks per lane");
};

template <typename KT, typename VT> struct Sc2SmallCfg {
	static constexpr int ELEM = sizeof(KT) > (size_t)val_bytes<VT>::value ? sizeof(KT) : val_bytes<VT>::value;
	static constexpr bool AVAILABLE = sizeof(KT) >= 4;   // (narrower keys: one 16-byte load would exceed the lane's keys)
	typedef Sc2Cfg<KT, VT, 16, 1, 8, false, AVAILABLE ? (ELEM == 8 ? 4 : 8) : 0> type;
};

// The staging area is addressed through a swizzle of the byte offset: bits 4-6 (which 16-byte unit of a 128-byte row) are
// XORed with bits 9-11.  A tile's runs lie one after the other in staging order, so when the digits of a column are spread
// perfectly evenly -- keys like i * c mod m: arithmetic sequences, strided indices, regular time stamps -- every run starts a
// multiple of 512 bytes after the previous one, the 64 lanes of a staging store then all hit ONE bank, and the staging phase
// takes 33 k cycles instead of 2.4 k (sawtooth).  With the swizzle such lanes spread over eight bank
// groups; 16-byte units stay whole (the write-out reads them with one instruction), random positions are as random as before.
// Keys-only passes only: in the key + payload passes the extra address arithmetic costs about 1 % (measured A/B on one box:
// cfg 4 3.16 against 3.14 ms), on keys-only passes nothing (headline 119.2 against 118.7 Gkeys/s, sawtooth 2.93 against 3.20 ms).
template <bool ON> __device__ __forceinline__ u32 stage_swz(u32 byte_off)
{
	return ON ? byte_off ^ (((byte_off >> 9) & 7u) << 4) : byte_off;
}

template <typename KT, typename VT, typename ST, typename C> struct Sc2Smem {
	__attribute__((aligned(16))) unsigned char stage_raw[C::STAGE_BYTES];
	u32 cell[C::TPS][C::NWAVES][C::CELL16 ? 128 : 256];   // cell per (tile, wave, digit): count, then run start / cursor
	ST delta[C::TPS][256];              // global offset of a digit's run minus its tile-local offset
	u32 wsum[C::TPS][4];
	u32 ticket;
	u32 smax;        // SCATTER_SELF_PLAN: the planned column's largest bin
	u64 soff[256];   // SCATTER_SELF_PLAN: the planned column's exclusive offsets (instead of gbase[])
};

// DIG_PLAIN: the key is its own KDF (unsigned, ascending): the digit is one
// bit-field extract.  DIG_GENERIC: kdf_apply, as in rsx_scatter_kernel.
// DIG_XOR: integer keys of either sign and order (no float mask): the KDF only complements fixed bits,
// so the digit is the plain bit field XOR one per-pass constant (0x80 on a signed key's top byte, 0xFF when descending).
enum { DIG_GENERIC = 0, DIG_PLAIN = 1, DIG_XOR = 2 };

template <int DIG, typename KT>
__device__ __forceinline__ u32 digit2(KT raw, const KdfArgs<KT> ka, u32 shift)
{
	if constexpr (DIG == DIG_PLAIN)
		return (u32)(raw >> shift) & 0xFFu;
	else if constexpr (DIG == DIG_XOR)
		return ((u32)(raw >> shift) & 0xFFu) ^ ((u32)((ka.sflip ^ ka.desc) >> shift) & 0xFFu);   // (the constant is scalar arithmetic)
	else
		return digit_of(raw, ka, shift);
}

// ---- hot digits (HOT kernels) -----------------------------------------------------------------------------------------
// An LDS atomic serialises the lanes of one instruction that hit the same address.  Where a byte column has a few
// dominant digits (small integers, floats of one magnitude, Zipf-like keys, columns with two or four values) 16 to 64
// lanes of every round hit one cell, for the counting and for the ranking atomic alike (2^28 keys with four values per
// byte: 0.80 ms per pass against 0.50 ms for uniform digits).  The HOT instantiation takes up to four hot digits of the
// column from rsx_plan_kernel (hotd[column]: digits holding a sixteenth of the keys or more) and, in whole tiles, ranks
// their lanes with one ballot per hot digit -- rank = the wave's running count of the digit + the number of lower lanes
// that have it -- so that only the lanes with other digits go to the LDS counters.  The ballots cost a few percent
// on uniform digits, so the host selects HOT per column, from the histogram it has anyway (`Plan::hot`: one digit holds
// an eighth of the keys or more).

// KTO: the type the keys are written out in.  A rank sort (rsx_sort_rank*) only needs the bytes of a key that later passes
// still look at: a pass writes kdf(key) >> oshift, narrowed to the smallest type that holds the columns to come, and the
// next pass reads that type with the identity KDF and its digit in the low byte (2^28 f32 keys -> u32 ranks: 60 -> 50 bytes
// of traffic per key).
// SEG: a pass INSIDE the level-1 buckets of a two-level sort (rsx_hybrid.hpp).  Tiles come from `seg.tiles` (cut at bucket
// boundaries, in bucket order = ticket order), a digit's offset is the bucket's start + the bucket's own exclusive scan
// (seg.hist, rsx_seg_plan_kernel), and the look-back chain ends at the bucket's first tile.  With SCATTER_SEG_LEAVES the pass
// goes by the level-2 column and only runs in SEG_MODE_LEAVES; without, it is pass `pass_index` of the LSB-first passes inside
// the buckets and only runs in SEG_MODE_LSD.
struct SegArgs {
	const SegTile *tiles;
	const SegCtl *ctl;
	const u32 *hist;   // [bucket][slot][256] exclusive offsets inside the bucket
	u32 slots;         // rows per bucket in `hist` (key bytes - 1)
	// SCATTER_SEG_SLACK: no counts are known; (digit, digit) bucket b * 256 + d is written into its own slot of slack_cap
	// keys of the scratch array `kout` points to, at the position the look-back chain alone gives (the keys of earlier tiles
	// of the bucket with the digit).  A slot that would overflow sets *overflow; the caller then discards the attempt.
	u32 slack_cap;
	u32 *overflow;
	// The level-1 slots of a keys-only sort without a histogram lie in TWO arrays (rsx.hip, blind_enqueue): slots 0 .. lo_slots-1
	// in the caller's second buffer (proven unsorted, the input leaves that buffer to the sort), the others in the library's
	// scratch array.  The level-1 pass is given ONE base (`kout`: the lower of the two arrays) and the element offsets of slot 0
	// of either part from it (out_off_lo for digits below lo_slots, out_off_hi -- of a virtual slot 0, lo_slots slots before the
	// scratch array -- for the others): they go into the digits' run offsets, so no store knows about it.  (Both within 2^32
	// elements of the base: the host checks.)  The level-2 pass reads buckets lo_slots .. from kin_hi (same element index).
	u32 out_off_lo, out_off_hi;
	const void *kin_hi;
	u32 lo_slots;
	// Key + payload and rank sorts: the payloads' level-1 slots lie in two arrays of their own (the caller's second payload buffer,
	// or the index buffer of a rank sort, and scratch), which need not lie as the keys' two do: their slot-0 offsets from `vout`
	// (level-1 pass) and the array of buckets lo_slots .. (level-2 pass), the same way.  The keys of a rank sort have no second
	// buffer: out_off_lo == out_off_hi, kin_hi null, while the indices' slots are split.
	u32 v_off_lo, v_off_hi;
	const void *vin_hi;
};

// SCATTER_SELF_PLAN (pass 0 of a blocking keys-only sort of a mid-size array): no plan kernel has run.  `gbase` is the
// histogram's RAW counts; every workgroup derives what it needs itself -- the kept columns from the bins of the first key's
// digits (radix_sort.hpp:64-70), the largest bin of the highest kept column and with it the hybrid decision (rsx_hybrid.hpp),
// the exclusive scan of the column it then sorts by (:72-80) -- a dozen scalar loads, 256 counts and one wave scan, while its
// tile's keys are not needed yet; workgroup 0 also writes the plan (device + the host's pinned copy) and, for the leaves, the
// scanned offsets.  A launch (and its gap) less: at 10^5 keys a sort is four launches of 5-7 us.
struct SelfPlanArgs {
	const u32 *unsorted;   // the histogram kernel's flag
	Plan *plan, *host_plan;
	u64 *gscan;            // [256] the highest kept column's exclusive offsets, when the plan is one MSB pass and leaves
	HybCaps caps;
};

template <typename KT, typename VT, typename ST, typename C = Sc2Cfg<KT, VT>, bool TL = false, int DIG = DIG_GENERIC,
          bool HOT_ = false, typename KTO = KT, bool SEG = false>
__global__ __launch_bounds__(C::BLOCK) void rsx_scatter2_kernel(const KT *__restrict__ kin, KTO *__restrict__ kout,
                                                                 const VT *__restrict__ vin, VT *__restrict__ vout, u64 n,
                                                                 u32 shift, const u64 *__restrict__ gbase, u32 tps, ST *status,
                                                                 u32 *ticket, KdfArgs<KT> ka, u32 flags, u64 *tl,
                                                                 const Plan *__restrict__ dplan = nullptr, u32 pass_index = 0,
                                                                 u32 oshift = 0, const u32 *__restrict__ hotd = nullptr,
                                                                 SegArgs seg = SegArgs{nullptr, nullptr, nullptr, 0, 0, nullptr, 0, 0, nullptr, 0},
                                                                 const void *__restrict__ kalt = nullptr,
                                                                 SelfPlanArgs sp = SelfPlanArgs{nullptr, nullptr, nullptr, nullptr,
                                                                                                HybCaps{0, 0, 0, 0}})
{
	constexpr bool NARROW = !std::is_same<KTO, KT>::value;
	// (SEG with KTO narrower than KT: only the pass into slots whose leaves need no more than KTO's bytes of the derived key --
	// SCATTER_SEG_SLACK, oshift 0; the in-place segmented passes exchange kin and kout and need one type)
	static_assert(!SEG || (C::TPS == 1 && !HOT_), "segmented passes: plain tiles");
	typedef StatusBits<ST> SB_;
	constexpr int NWAVES = C::NWAVES, BLOCK = C::BLOCK, KPT = C::KPT, TPS = C::TPS, SB = C::SB, CHUNK = C::CHUNK;
	// Device-scheduled pass: launched before the host has seen the plan (the first pass of every sort, so that the host's
	// wait for the plan is hidden behind it; every pass of rsx_sort_inplace_async, which never waits).  The pass is the
	// `pass_index`-th kept column of the device-side plan: it finds its column there, reads from the second buffer when
	// its index is odd, and does nothing if the input is sorted (radix_sort.hpp:60-62: `aux` must stay untouched then)
	// or has fewer kept columns.  `gbase` is the histogram's column 0 in this case.
	u32 dcol = 0;
	u32 seg_slot = 0;
	KT bl_cmask = 0, bl_key0 = 0;   // SCATTER_BLIND_TOP: the columns taken for constant and the first key (derived): checked on every key
	// SegCtl::compact (rank sorts of 4-byte keys whose varying bits are few, rsx_hybrid.hpp): the level-1 pass packs them and every
	// pass behind it sorts the packed keys -- plain unsigned keys, whatever the caller's type
	constexpr bool CAN_COMPACT = SEG && sizeof(KT) == 4 && val_bytes<VT>::value == 4 && std::is_same<KTO, KT>::value;
	// (... and the level-2 pass that writes the low two bytes of what it reads -- KTO = u16, pairs_blind_enqueue -- reads packed keys too)
	constexpr bool SEES_COMPACT = SEG && sizeof(KT) == 4 && val_bytes<VT>::value == 4;
	u32 cp_nb = 0, cp_raw0 = 0, cp_vnot = 0, cp_kind = 0;
	u32 cp_piece[4] = {0, 0, 0, 0};
	const KdfArgs<KT> ka_raw = ka;
	if constexpr (SEG) {
		if ((flags & SCATTER_BLIND) && seg.ctl->blind != BLIND_GO)
			return;   // (a sort without a histogram that has been called off: rsx_hybrid.hpp)
		const u32 mode = seg.ctl->mode;
		if constexpr (SEES_COMPACT) {
			if ((flags & SCATTER_BLIND) && seg.ctl->compact) {
				cp_nb = seg.ctl->compact;
				if (flags & SCATTER_BLIND_TOP) {
					cp_raw0 = seg.ctl->craw0;
					cp_vnot = seg.ctl->cvnot;
					cp_kind = seg.ctl->ckind;
#pragma unroll
					for (int i = 0; i < 4; ++i)
						cp_piece[i] = seg.ctl->cpiece[i];
				}
				ka.fmask = ka.sflip = ka.desc = 0;   // (what is ranked from here on is a packed key: its own derived key)
			}
		}
		if (flags & SCATTER_BLIND_TOP) {
			// the level-1 pass of such a sort: by the highest column the sample proved kept, no offsets
			seg_slot = 0;
			shift = seg.ctl->shift1;   // (8 x the highest kept column, or the bits below the highest varying one: rsx_blind_precheck_kernel)
			bl_cmask = (KT)(((u64)seg.ctl->cmask_hi << 32) | seg.ctl->cmask_lo);
			bl_key0 = (KT)(((u64)seg.ctl->key0_hi << 32) | seg.ctl->key0_lo);
		} else if (dplan->hyb != HYB_TWO_LEVEL) {
			return;
		} else if (flags & SCATTER_SEG_SLACK) {    // the pass by the level-2 column into slots of the scratch array, before anything is decided
			seg_slot = dplan->ncols - 2;
		} else if (flags & SCATTER_SEG_LEAVES) {   // the pass by the level-2 column, aux -> src; enqueued before the host knows the mode
			if (mode != SEG_MODE_LEAVES)
				return;
			seg_slot = dplan->ncols - 2;
		} else {                            // LSB-first pass number pass_index inside the buckets (even: aux -> src)
			if (mode != SEG_MODE_LSD || pass_index >= dplan->ncols - 1)
				return;
			seg_slot = pass_index;
			if constexpr (!NARROW) {
				if (pass_index & 1) {
					const KT *t = kin;
					kin = (const KT *)kout;
					kout = (KTO *)const_cast<KT *>(t);
				}
			}
		}
		if constexpr (sizeof(KT) == 8) {
			// 8-byte keys, the level-2 pass of a sort without a histogram: the sample decided whether the slots hold whole keys or
			// the low word of the derived keys (SegCtl::narrow); both forms are enqueued, the other one leaves
			if ((flags & SCATTER_BLIND) && (flags & SCATTER_SEG_SLACK) && !(flags & SCATTER_BLIND_TOP) &&
			    (seg.ctl->narrow != 0u) != NARROW)
				return;
		}
		if (!(flags & SCATTER_BLIND_TOP)) {
			shift = ((flags & SCATTER_BLIND) && (flags & SCATTER_SEG_SLACK)) ? seg.ctl->shift2 : 8 * dplan->cols[seg_slot];
			gbase += 256 * dplan->cols[dplan->ncols - 1];   // the level-1 column's offsets: bucket starts
		}
	} else if (dplan && !(flags & SCATTER_SELF_PLAN)) {
		// (rsx_sort_inplace_async, after an attempt without the histogram that went through: the keys are sorted already)
		if (seg.ctl && seg.ctl->mode == SEG_MODE_LEAVES)
			return;
		if (dplan->sorted || pass_index >= dplan->ncols)
			return;
		if ((flags & SCATTER_ONE_COL_FILLED) && dplan->ncols == 1)
			return;   // keys only, one kept column: the sorted array is written from the histogram (rsx_fill_runs_kernel)
		if (dplan->hyb && pass_index != 0)
			return;   // (one MSB pass and leaves, rsx_hybrid.hpp: only pass 0 is device-scheduled)
		// pass 0 of a hybrid sort goes by the HIGHEST kept column (README.md:647-650)
		const u32 col = dplan->cols[dplan->hyb ? dplan->ncols - 1 : pass_index];
		dcol = col;
		shift = 8 * col;
		gbase += 256 * col;
		if constexpr (!NARROW) {
			if (flags & SCATTER_RANK_ASYNC) {
				// rsx_sort_rank_inplace_async: the caller's keys are read by pass 0 only (which also makes the indices), the
				// work copies kout / kalt alternate after that, the last pass writes no keys, and the halves of the index
				// buffer are taken so that the LAST pass writes the first one, however many passes the plan has.
				const u32 P = dplan->ncols, i = pass_index;
				KT *k0 = kout, *k1 = (KT *)const_cast<void *>(kalt);
				kin = i == 0 ? kin : (((i - 1) & 1u) ? k1 : k0);
				kout = (i & 1u) ? k1 : k0;
				const u32 w = (P - 1 - i) & 1u;
				VT *h0 = const_cast<VT *>(vin), *h1 = vout;
				vout = w ? h1 : h0;
				vin = w ? h0 : h1;
				if (i == 0)
					flags |= SCATTER_GEN_INDEX;
				if (i == P - 1)
					flags |= SCATTER_SKIP_KEYS;
			} else if (pass_index & 1) {
				const KT *t = kin;
				kin = kout;
				kout = const_cast<KT *>(t);
				const VT *u = vin;
				vin = vout;
				vout = const_cast<VT *>(u);
			}
		}
	}
	constexpr bool HAS_VAL = val_bytes<VT>::value != 0;
	// One tile per super-tile: the tile's keys stay in registers between the count and the staging, so they
	// are read from memory once.  (With more tiles per super-tile they are re-read out of L2 / Infinity Cache.)
	constexpr bool KEEP = TPS == 1;
	constexpr bool HOT = HOT_ && !C::CELL16;
	static_assert(!HOT || TPS == 1, "the hot digits' counts are kept per tile in registers");
	// HOT: up to four digits of this column that hold a sixteenth of the keys or more (rsx_plan_kernel, hotd[column]).
	// In whole tiles they never touch the LDS counters: a ballot per hot digit gives a lane its rank among the wave's
	// lanes with that digit, the counts and the cursors of the wave's hot digits live in (uniform) registers.
	constexpr int NHOT = 4;
	u32 hk[NHOT] = {0, 0, 0, 0}, hc[NHOT] = {0, 0, 0, 0};
	bool hv[NHOT] = {false, false, false, false};
	if constexpr (HOT) {
		const u32 hcol = dplan ? dcol : (flags >> SCATTER_COL_SHIFT) & 7u;
		const u32 hw = hotd ? __builtin_amdgcn_readfirstlane(hotd[hcol]) : 0u;
		const u32 hb = hotd ? __builtin_amdgcn_readfirstlane(hotd[8]) >> (4 * hcol) : 0u;
#pragma unroll
		for (int k = 0; k < NHOT; ++k) {
			hk[k] = (hw >> (8 * k)) & 0xFFu;
			hv[k] = ((hb >> k) & 1u) != 0;
		}
	}
	KT keep[KEEP ? KPT : 1];
	// RANK1: ONE returning LDS atomic per key.  The count phase's atomic already is the key's rank inside its (wave, digit)
	// run (rounds in memory order, lanes in lane order); it is kept in 16 bits, and after the layout the staging position is
	// the run's start + rank -- an LDS read where a second atomic used to be.  On uniform digits the pass is as long as with
	// two atomics (the phase ends with the chain either way), on skewed digits, where lanes queue at one cell, 25-30 % shorter
	// (tools/ubench/rsx_scatter10_one_atomic.hpp).
	// (keys-only passes: in the pair passes sixteen more registers live across the layout make several instantiations spill;
	// round 3 tried it there again WITHOUT the early payload requests, which frees 32 registers: 52-76 bytes of scratch per
	// lane remain and 2^28 f32 keys -> ranks take 3.16 / 3.30 / 3.42 ms against 3.10 / 3.39 / 3.32, pairs 4.49 against 4.26)
	constexpr bool RANK1 = KEEP && !C::CELL16 && !HAS_VAL;
	u32 rk[RANK1 ? KPT / 2 : 1];
	if constexpr (RANK1) {
#pragma unroll
		for (int i = 0; i < KPT / 2; ++i)
			rk[i] = 0;
	}
	__shared__ Sc2Smem<KT, VT, ST, C> sm;
	const u32 tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
	const u64 t_start = TL ? __builtin_readcyclecounter() : 0;
	bool self_planned = false;
	if constexpr (!SEG && !NARROW && !HAS_VAL && !HOT_) {
		if (flags & SCATTER_SELF_PLAN) {
			constexpr u32 WC = sizeof(KT);
			// everything that comes from memory is requested at once: the first key, the sorted flag, and every column's
			// 256 counts (thread d < 256 holds digit d's) -- one round trip instead of three dependent ones
			const KT key0raw = kin[0];
			const u32 unsorted_flag = *sp.unsorted;
			u64 cnts[WC];
#pragma unroll
			for (u32 c = 0; c < WC; ++c)
				cnts[c] = tid < 256 ? gbase[256 * c + tid] : 0;
			const KT key0 = kdf_apply(key0raw, ka);                     // radix_sort.hpp:65
			u32 *const skept = (u32 *)&sm.cell[0][0][0] + 256;          // (scratch: the cells are zeroed further down)
#pragma unroll
			for (u32 c = 0; c < WC; ++c)
				if (tid == ((u32)(key0 >> (8 * c)) & 0xFFu))
					skept[c] = cnts[c] != n ? 1u : 0u;                  // :67
			if (tid == 0) {
				sm.smax = 0;
				skept[8] = 0;
			}
			__syncthreads();
			u32 nc = 0;
			u64 colpack = 0;                                            // 4 bits per kept column, LSB first (:66-69)
#pragma unroll
			for (u32 c = 0; c < WC; ++c)
				if (skept[c]) {
					colpack |= (u64)c << (4 * nc);
					++nc;
				}
			const bool sorted = unsorted_flag == 0;                     // :60
			const u32 top = nc ? (u32)(colpack >> (4 * (nc - 1))) & 15u : 0u;
			u64 cnt_top = 0;
#pragma unroll
			for (u32 c = 0; c < WC; ++c)
				if (c == top)
					cnt_top = cnts[c];
			if (tid < 256)
				atomicMax(&sm.smax, cnt_top > 0xFFFFFFFFull ? 0xFFFFFFFFu : (u32)cnt_top);
			// a dominant digit (an eighth of the keys or more, Plan::hot) in a column the leaves would sort by: no leaves
			bool mine_hot = false;
#pragma unroll
			for (u32 c = 0; c < WC; ++c)
				mine_hot |= c != top && skept[c] && cnts[c] >= n / 8 + 1;
			if (mine_hot)
				skept[8] = 1u;
			__syncthreads();
			const u32 max1 = nc ? sm.smax : 0u;
			const u32 hyb = (!sorted && n < (1ull << 30) && sp.caps.cap1 && nc >= sp.caps.min_cols1 && max1 <= sp.caps.cap1 && !skept[8]) ? 1u : 0u;
			const u32 col = hyb ? top : (u32)colpack & 15u;              // one MSB pass and leaves go by the HIGHEST kept column
			u64 cnt = 0;
#pragma unroll
			for (u32 c = 0; c < WC; ++c)
				if (c == col)
					cnt = cnts[c];
			__syncthreads();                                            // (skept has been read by everybody)
			if (tid < 256)
				sm.soff[tid] = cnt;
			__syncthreads();
			if (tid < 64)
				wave_scan_256(sm.soff, (u64 *)&sm.cell[0][0][0], tid);  // :72-80
			__syncthreads();
			if (blockIdx.x == 0) {
				if (hyb && tid < 256)
					sp.gscan[tid] = sm.soff[tid];
				if (tid == 0) {
					Plan *const out[2] = {sp.plan, sp.host_plan};
#pragma unroll
					for (int k = 0; k < 2; ++k) {
						out[k]->ncols = nc;
						out[k]->sorted = sorted ? 1u : 0u;
						for (u32 i = 0; i < 8; ++i)
							out[k]->cols[i] = i < nc ? (u32)(colpack >> (4 * i)) & 15u : 0u;
						out[k]->hot = 0;
						out[k]->vary_lo = out[k]->vary_hi = 0;
						out[k]->hyb = hyb;
						out[k]->max1 = max1;
					}
					__threadfence_system();
				}
			}
			if (sorted || nc == 0)
				return;                                                 // :60-62: `aux` stays untouched
			if ((flags & SCATTER_ONE_COL_FILLED) && nc == 1)
				return;
			dcol = col;
			shift = 8 * col;
			self_planned = true;
		}
	}

	if (tid == 0) {
		sm.ticket = atomicAdd(ticket, 1u);   // super-tiles are handed out in start order => look-back cannot deadlock
		if constexpr (SEG)   // (blind level-1 pass: has a slot overflowed already?  one answer for the whole workgroup)
			sm.smax = (flags & SCATTER_BLIND_TOP) ? __hip_atomic_load(seg.overflow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
	}
	constexpr int CW = C::CELL16 ? 128 : 256;   // words per (tile, wave)
	for (u32 i = tid; i < TPS * NWAVES * CW; i += BLOCK)
		(&sm.cell[0][0][0])[i] = 0;
	__syncthreads();
	u32 stile = __builtin_amdgcn_readfirstlane(sm.ticket);
	if (flags & SCATTER_DBG_XCD_RUNS) {
		// probe only (grid a multiple of 8 << lr, lr <= 5): workgroup i runs on XCD i % 8 and takes its k = i / 8 -th tile
		// there, tiles dealt to the XCDs in runs of 2^lr.  Measured: no faster than tickets.
		const u32 lr = (flags >> SCATTER_XCD_RUN_SHIFT) & 15u, R = 1u << lr;
		const u32 x = blockIdx.x & 7u, k = blockIdx.x >> 3;
		stile = ((k >> lr) * 8u + x) * R + (k & (R - 1u));
	}
	u64 beg = (u64)stile * tps * C::TILE;
	u64 end = beg + (u64)tps * C::TILE;
	if (end > n)
		end = n;
	u32 seg_first = 0, seg_bucket = 0;
	if constexpr (SEG) {
		if (flags & SCATTER_BLIND_TOP) {
			// plain tiles, one bucket (the whole array), the chain ends at tile 0
			if (beg >= n)
				return;
			if (sm.smax) {
				// the attempt is lost (a slot overflowed): pass the chain on and leave -- what this pass writes is discarded
				if (tid < 256)
					__hip_atomic_store(status + (stile * 256u + tid), (ST)ST_PREFIX << SB_::SHIFT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				return;
			}
		} else {
			if (stile >= seg.ctl->ntiles)
				return;
			const SegTile st = seg.tiles[stile];
			beg = st.beg;
			end = beg + st.cnt;
			seg_first = st.first;
			seg_bucket = st.bucket;
		}
	}
	u64 vdiff = 0;                                // (SEG with payloads: see SegArgs::vin_hi, v_off_*)
	u32 vadj_lo = 0, vadj_hi = 0, vadj_n = 0;
	if constexpr (SEG) {
		// SegArgs::kin_hi: the tile lies in the other array -- the same element index from there, as a shift of the tile's bounds.
		// (Where this stands matters to the compiler: the same lines next to the tile-table read above, or an assignment to
		// `kin` there, gave a build whose level-2 pass faulted even with kin_hi null.)
		u64 shift_elems = 0;
		if (seg.kin_hi && seg_bucket >= seg.lo_slots)
			shift_elems = (u64)(((const char *)seg.kin_hi - (const char *)kin) / (long long)sizeof(KT));
		beg += shift_elems;
		end += shift_elems;
		if constexpr (HAS_VAL) {
			// SegArgs::vin_hi: the payloads of this tile, as a difference to the keys' element index (modulo 2^64)
			if (seg.vin_hi && seg_bucket >= seg.lo_slots)
				vdiff = (u64)(((const char *)seg.vin_hi - (const char *)vin) / (long long)sizeof(VT)) - shift_elems;
			if (flags & SCATTER_BLIND_TOP) {
				vadj_lo = seg.v_off_lo - seg.out_off_lo;
				vadj_hi = seg.v_off_hi - seg.out_off_hi;
				vadj_n = seg.lo_slots;
			}
		}
	}
	const u32 wofs = wid * (64 * KPT) + lane;   // wave w owns [w*64*KPT, +64*KPT) of a tile; round r: element 64 r + lane
	// Element `base + wofs + 64 r` of an array: a uniform (scalar) address for (base, r) plus ONE 32-bit lane offset,
	// instead of a 64-bit address per round kept in registers (or spilled) across the tile.
	auto elem = [&](const auto *arr, const u64 base, const int r, const u32 wo) {
		typedef std::remove_cv_t<std::remove_pointer_t<decltype(arr)>> T;
		return *(const T *)((const char *)(arr + base + (u64)r * 64) + wo * (u32)sizeof(T));
	};
	// A partial tile (the last one) works on an opaque copy of wofs: what it derives from it (32 bounds tests, ...) is
	// then computed there and not hoisted in front of the branch, where every tile would pay for it with registers.
	auto opaque = [](u32 x) {
		asm volatile("" : "+v"(x));
		return x;
	};

	// SCATTER_UNSTABLE (keys-only passes into buckets that leaves sort): one row of cells for the whole workgroup
	const bool unstable = RANK1 && !C::CELL16 && (flags & SCATTER_UNSTABLE) != 0;
	// ---- phase A: count, per tile and wave.  Order inside a wave's slice does not matter here, so the
	// slice is streamed with 16-byte loads (when the keys are 16-byte aligned), all of them in flight.
	constexpr int VEC = 16 / sizeof(KT);
	const bool vec_ok = !SEG && (((uintptr_t)kin) & 15) == 0 && !(flags & SCATTER_ELEM_LOADS);   // (a bucket starts anywhere)
#pragma unroll
	for (int t = 0; t < TPS; ++t) {
		const u64 base = beg + (u64)t * C::TILE;
		if (t < (int)tps && base < end) {
			const u32 cnt = (end - base) < (u64)C::TILE ? (u32)(end - base) : (u32)C::TILE;
			u32 *wc = sm.cell[t][unstable ? 0u : wid];
			if (vec_ok && cnt == (u32)C::TILE) {
				typedef KT vec_t __attribute__((ext_vector_type(VEC)));
				constexpr int NV = KPT / VEC;   // 16-byte loads per lane
				const vec_t *vp = (const vec_t *)(kin + base + (u64)wid * (64 * KPT)) + lane;
				vec_t v[NV];
#pragma unroll
				for (int i = 0; i < NV; ++i)
					v[i] = vp[i * 64];
				if constexpr (KEEP) {
					// 16-byte loads give lane l the elements 4(64 i + l) .. +3; ranking wants round r = element 64 r + l.
					// Transpose through the wave's own slice of the (still unused) staging area: linear write, strided
					// read.  DS operations of one wave execute in issue order, so no barrier is needed.
					// Vector i holds the rounds VEC i .. VEC i + VEC - 1: they are read back and counted as soon as it has
					// arrived, while the later vectors are still on their way.
					KT *scratch = (KT *)sm.stage_raw + (u32)wid * (64 * KPT);
#pragma unroll
					for (int i = 0; i < NV; ++i) {
						*((vec_t *)scratch + i * 64 + lane) = v[i];
						RSX_COMPILER_FENCE();
#pragma unroll
						for (int r = i * VEC; r < (i + 1) * VEC; ++r)
							keep[r] = scratch[r * 64 + lane];
#pragma unroll
						for (int r = i * VEC; r < (i + 1) * VEC; ++r) {
							const u32 d = digit2<DIG>(keep[r], ka, shift);
							if constexpr (C::CELL16) {
								atomicAdd(&wc[d >> 1], 1u << ((d & 1u) * 16u));
							} else if constexpr (HOT) {
								// a hot digit's rank: the wave's count of it so far + the lower lanes of this round that have it
								bool mine = false;
								u32 rank = 0;
#pragma unroll
								for (int k = 0; k < NHOT; ++k) {
									if (hv[k]) {
										const bool is = d == hk[k];
										const u64 m = __ballot(is);
										if (is)
											rank = hc[k] + mbcnt64(m);
										hc[k] += (u32)__popcll(m);
										mine |= is;
									}
								}
								if constexpr (RANK1) {
									if (!mine)
										rank = __hip_atomic_fetch_add(&wc[d], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
									rk[r >> 1] |= rank << (16 * (r & 1));
								} else if (!mine) {
									atomicAdd(&wc[d], 1u);
								}
							} else if constexpr (RANK1) {
								const u32 rank = __hip_atomic_fetch_add(&wc[d], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
								rk[r >> 1] |= rank << (16 * (r & 1));
							} else {
								atomicAdd(&wc[d], 1u);
							}
						}
						RSX_COMPILER_FENCE();
					}
					if constexpr (HOT) {   // the wave's counts of the hot digits (nobody else touches these cells)
#pragma unroll
						for (int k = 0; k < NHOT; ++k)
							if (hv[k] && lane == (u32)k)
								wc[hk[k]] = hc[k];
					}
				} else {
#pragma unroll
					for (int i = 0; i < NV; ++i) {
#pragma unroll
						for (int e = 0; e < VEC; ++e) {
							const u32 d = digit2<DIG>((KT)v[i][e], ka, shift);
							if constexpr (C::CELL16)
								atomicAdd(&wc[d >> 1], 1u << ((d & 1u) * 16u));
							else
								atomicAdd(&wc[d], 1u);
						}
					}
				}
			} else if (KEEP && cnt == (u32)C::TILE) {
				// whole tile, element loads: lane l of round r loads element 64 r + l itself (a wave-instruction reads 64
				// consecutive keys): four times the load instructions of the vector path, but no transposition through the LDS
				if constexpr (KEEP) {
					const KT *p = kin + base + wofs;
#pragma unroll
					for (int r = 0; r < KPT; ++r)
						keep[r] = p[r * 64];
					if constexpr (SEG) {
						if (bl_cmask) {   // (uniform) a key that differs from the first one in a column taken for constant ends the attempt
							KT bad = 0;
#pragma unroll
							for (int r = 0; r < KPT; ++r)
								bad |= ((DIG == DIG_PLAIN ? keep[r] : kdf_apply(keep[r], ka)) ^ bl_key0) & bl_cmask;
							if (__ballot(bad != 0) && lane == 0)
								atomicOr(seg.overflow, 1u);
						}
					}
					if constexpr (CAN_COMPACT) {
						if (cp_nb && (flags & SCATTER_BLIND_TOP)) {   // (uniform) the keys as the caller wrote them -> packed keys
							u32 bad = 0;
#pragma unroll
							for (int r = 0; r < KPT; ++r) {
								if (cp_kind) {   // (uniform) floats on a grid: the key as a fixed-point integer
									keep[r] = (KT)fixedpoint_key((u32)keep[r], cp_piece[0], cp_nb, cp_piece[1] != 0, ka_raw.desc != 0, bad);
								} else {
									bad |= ((u32)keep[r] ^ cp_raw0) & cp_vnot;
									keep[r] = (KT)compact_key<KT>((u32)keep[r], cp_piece, cp_nb, ka_raw);
								}
							}
							if (__ballot(bad != 0) && lane == 0)
								atomicOr(seg.overflow, 1u);   // (a bit the sample took for constant varies: the attempt is lost)
						}
					}
#pragma unroll
					for (int r = 0; r < KPT; ++r) {
						const u32 d = digit2<DIG>(keep[r], ka, shift);
						if constexpr (C::CELL16) {
							atomicAdd(&wc[d >> 1], 1u << ((d & 1u) * 16u));
						} else if constexpr (RANK1) {
							const u32 rank = __hip_atomic_fetch_add(&wc[d], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
							rk[r >> 1] |= rank << (16 * (r & 1));
						} else {
							atomicAdd(&wc[d], 1u);
						}
					}
				}
			} else if constexpr (KEEP) {
				const u32 wo = opaque(wofs);
#pragma unroll
				for (int r = 0; r < KPT; ++r) {
					const u32 o = wo + r * 64;
					keep[r] = o < cnt ? elem(kin, base, r, wo) : (KT)0;
				}
				if constexpr (SEG) {
					if (bl_cmask) {
						KT bad = 0;
#pragma unroll
						for (int r = 0; r < KPT; ++r)
							if (wo + r * 64 < cnt)
								bad |= ((DIG == DIG_PLAIN ? keep[r] : kdf_apply(keep[r], ka)) ^ bl_key0) & bl_cmask;
						if (__ballot(bad != 0) && lane == 0)
							atomicOr(seg.overflow, 1u);
					}
				}
				if constexpr (CAN_COMPACT) {
					if (cp_nb && (flags & SCATTER_BLIND_TOP)) {
						u32 bad = 0;
#pragma unroll
						for (int r = 0; r < KPT; ++r) {
							if (cp_kind) {
								u32 b2 = 0;
								keep[r] = (KT)fixedpoint_key((u32)keep[r], cp_piece[0], cp_nb, cp_piece[1] != 0, ka_raw.desc != 0, b2);
								if (wo + r * 64 < cnt)
									bad |= b2;
							} else {
								if (wo + r * 64 < cnt)
									bad |= ((u32)keep[r] ^ cp_raw0) & cp_vnot;
								keep[r] = (KT)compact_key<KT>((u32)keep[r], cp_piece, cp_nb, ka_raw);
							}
						}
						if (__ballot(bad != 0) && lane == 0)
							atomicOr(seg.overflow, 1u);
					}
				}
#pragma unroll
				for (int r = 0; r < KPT; ++r) {
					const u32 o = wo + r * 64;
					if (o < cnt) {
						const u32 d = digit2<DIG>(keep[r], ka, shift);
						if constexpr (C::CELL16) {
							atomicAdd(&wc[d >> 1], 1u << ((d & 1u) * 16u));
						} else if constexpr (RANK1) {
							const u32 rank = __hip_atomic_fetch_add(&wc[d], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
							rk[r >> 1] |= rank << (16 * (r & 1));
						} else {
							atomicAdd(&wc[d], 1u);
						}
					}
				}
			} else {
				const u32 wo = opaque(wofs);
#pragma unroll 1
				for (int r0 = 0; r0 < KPT; r0 += SB) {
					KT cur[SB];
#pragma unroll
					for (int r = 0; r < SB; ++r) {
						const u32 o = wo + (r0 + r) * 64;
						cur[r] = o < cnt ? elem(kin, base, r0 + r, wo) : (KT)0;
					}
#pragma unroll
					for (int r = 0; r < SB; ++r) {
						const u32 o = wo + (r0 + r) * 64;
						if (o < cnt) {
							const u32 d = digit2<DIG>(cur[r], ka, shift);
							if constexpr (C::CELL16)
								atomicAdd(&wc[d >> 1], 1u << ((d & 1u) * 16u));
							else
								atomicAdd(&wc[d], 1u);
						}
					}
				}
			}
		}
	}
	__syncthreads();
	if (TL && tid == 0)
		tl[(u64)stile * 16 + 1] = __builtin_readcyclecounter();

	// ---- digit thread d: totals, publish the aggregate, START the look-back, tile layouts; the chain is
	// resolved (and the prefix published) after the layout, while waves 4.. are already staging their keys.
	unsigned short *cell16 = (unsigned short *)&sm.cell[0][0][0];   // [TPS][NWAVES][256]
	constexpr int LB = C::LB;
	u32 tc[TPS], incl[TPS], tb[TPS];
	ST w[LB];
	u32 st_cnt = 0;
	int back = (int)stile - 1;   // nearest predecessor not consumed yet
	ST *my_status = status + (stile * 256u + tid);   // (32-bit element offsets from the uniform base)
	// LB predecessors are fetched per round trip (independent loads) and consumed in order
	auto look = [&]() {
		const u32 t = opaque(tid);   // (or the compiler keeps LB loop-invariant offsets in registers)
#pragma unroll
		for (int j = 0; j < LB; ++j) {
			const int lo = SEG ? (int)seg_first : 0;
			const int p = back - j > lo ? back - j : lo;   // the (bucket's) first tile always holds a prefix: safe filler
			w[j] = __hip_atomic_load(status + ((u32)p * 256u + t), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		}
	};
	if (tid < 256) {
#pragma unroll
		for (int t = 0; t < TPS; ++t) {
			u32 c = 0;
			if (unstable) {
				c = sm.cell[t][0][tid & (CW - 1)];
			} else {
#pragma unroll
				for (int k = 0; k < NWAVES; ++k)
					c += C::CELL16 ? (u32)cell16[(t * NWAVES + k) * 256 + tid] : sm.cell[t][k][tid & (CW - 1)];
			}
			tc[t] = c;
			st_cnt += c;
		}
		const ST word = ((ST)(stile == (SEG ? seg_first : 0u) ? ST_PREFIX : ST_AGGREGATE) << SB_::SHIFT) | (ST)st_cnt;
		__hip_atomic_store(my_status, word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		if (stile != (SEG ? seg_first : 0u))
			look();
		// prefix over digits, per tile (wave scan now, wave totals through LDS)
#pragma unroll
		for (int t = 0; t < TPS; ++t) {
			u32 x = tc[t];
#pragma unroll
			for (int off = 1; off < 64; off <<= 1) {
				const u32 y = __shfl_up(x, off);
				if (lane >= (u32)off)
					x += y;
			}
			incl[t] = x;
			if (lane == 63)
				sm.wsum[t][opaque(wid)] = x;
		}
	}
	__syncthreads();
	if (tid < 256) {
#pragma unroll
		for (int t = 0; t < TPS; ++t) {
			u32 tbase = incl[t] - tc[t];
			for (u32 k = 0; k < wid; ++k)
				tbase += sm.wsum[t][k];
			tb[t] = tbase;
			u32 acc = tbase;   // counts -> run starts, in place
			if (unstable) {
				sm.cell[t][0][tid & (CW - 1)] = tbase;
				continue;
			}
#pragma unroll
			for (int k = 0; k < NWAVES; ++k) {
				u32 c;
				if constexpr (C::CELL16) {
					c = cell16[(t * NWAVES + k) * 256 + tid];
					cell16[(t * NWAVES + k) * 256 + tid] = (unsigned short)acc;
				} else {
					c = sm.cell[t][k][tid & (CW - 1)];
					sm.cell[t][k][tid & (CW - 1)] = acc;
				}
				acc += c;
			}
		}
	}
	__syncthreads();
	if (TL && tid == 0) {
		tl[(u64)stile * 16 + 0] = t_start;
		tl[(u64)stile * 16 + 2] = __builtin_readcyclecounter();
	}
	if (tid < 256) {
		// the chain: aggregates are summed until the first inclusive prefix; an empty word ends the batch
		u64 excl = 0;
		u32 depth = 0;
		if (stile != (SEG ? seg_first : 0u)) {
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
			const ST pword = ((ST)ST_PREFIX << SB_::SHIFT) | (ST)(excl + st_cnt);
			__hip_atomic_store(my_status, pword, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		}
		u64 running;
		if constexpr (SEG) {
			if (flags & SCATTER_SEG_SLACK) {
				running = ((u64)seg_bucket * 256 + tid) * seg.slack_cap + excl;
				if (excl + st_cnt > (u64)seg.slack_cap) {
					// the slot is too small: the attempt will be discarded (rsx_seg_slack_plan_kernel sees the flag); this run
					// goes to the dump area behind the last slot (a tile of padding), wherever the chain would have put it --
					// a heavy bucket's runs must not walk over the end of the scratch array
					atomicOr(seg.overflow, 1u);
					running = (u64)((flags & SCATTER_BLIND_TOP) ? 256u : 65536u) * seg.slack_cap;
					// (slots in the caller's buffer, which has no room behind them: over the slot's own beginning -- a level-1
					// slot holds more than a tile there, blind_enqueue)
					if ((flags & SCATTER_BLIND_TOP) && tid < seg.lo_slots)
						running = (u64)tid * seg.slack_cap;
				}
				if (flags & SCATTER_BLIND_TOP)   // (SegArgs::out_off_*: which array the digit's slot lies in)
					running += tid < seg.lo_slots ? seg.out_off_lo : seg.out_off_hi;
			} else {
				running = gbase[seg_bucket] + seg.hist[((u64)seg_bucket * seg.slots + seg_slot) * 256 + tid] + excl;
			}
		} else {
			running = (self_planned ? sm.soff[tid] : gbase[tid]) + excl;
		}
#pragma unroll
		for (int t = 0; t < TPS; ++t) {
			sm.delta[t][tid] = (ST)(running - tb[t]);   // modulo 2^32 when ST is 32-bit (n < 2^30 then)
			running += tc[t];
		}
		if (TL && tid == 0) {
			tl[(u64)stile * 16 + 3] = __builtin_readcyclecounter();
			tl[(u64)stile * 16 + 12] = depth;
		}
	}

	// ---- phase B: the tiles, in order
	auto do_tile = [&](auto full_c, const int t, const u64 base, const u32 cnt) {
		constexpr bool full = decltype(full_c)::value;   // a whole tile: no bounds checks
		const u32 wo = full ? wofs : opaque(wofs);
		u32 *wc = sm.cell[t][unstable ? 0u : wid];
		const ST *delta = sm.delta[t];
		u32 hcur[NHOT] = {0, 0, 0, 0};   // HOT: the wave's cursors of the hot digits = their run starts after the layout
		if constexpr (HOT && !RANK1) {
			if (full) {
#pragma unroll
				for (int k = 0; k < NHOT; ++k)
					if (hv[k])
						hcur[k] = (u32)__builtin_amdgcn_readfirstlane((int)wc[hk[k]]);
			}
		}

		// rank + stage: the returning atomic on the (wave, digit) cursor is the key's tile-local position.
		// Rounds are issued in memory order; lanes of a round come back in lane order (see the header).
		u32 posp[(HAS_VAL && !RANK1) ? KPT / 2 : 1];   // (RANK1: a key's staged position replaces its rank in rk[])
		if constexpr (HAS_VAL) {
#pragma unroll
			for (int i = 0; i < KPT / 2; ++i)
				if constexpr (!RANK1)
					posp[i] = 0;
		}
		auto load_batch = [&](KT (&dst)[SB], const int r0) {
#pragma unroll
			for (int r = 0; r < SB; ++r) {
				const u32 o = wo + (r0 + r) * 64;
				if (TL && (flags & SCATTER_DBG_NOLOADB))
					dst[r] = (KT)((o ^ (u32)base) * 2654435761u);
				else
					dst[r] = (full || o < cnt) ? elem(kin, base, r0 + r, wo) : (KT)0;
			}
		};
		auto stage_batch = [&](const KT (&cur)[SB], const int r0) {
			// all the batch's atomics are issued before the first position is needed: the returning
			// atomics of a wave pipeline in the LDS; only then the keys are stored at their positions
			u32 pos[SB];
#pragma unroll
			for (int r = 0; r < SB; ++r) {
				const u32 o = wo + (r0 + r) * 64;
				pos[r] = 0;
				if (full || o < cnt) {
					const u32 d = digit2<DIG>(cur[r], ka, shift);
					if constexpr (RANK1) {
						pos[r] = wc[d] + ((rk[(r0 + r) >> 1] >> (16 * ((r0 + r) & 1))) & 0xFFFFu);
					} else if constexpr (C::CELL16) {
						const u32 sh = (d & 1u) * 16u;
						const u32 old = __hip_atomic_fetch_add(&wc[d >> 1], 1u << sh, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
						pos[r] = (old >> sh) & 0xFFFFu;
					} else if constexpr (HOT) {
						bool mine = false;
						if (full) {   // (whole tiles only: every lane is active there)
#pragma unroll
							for (int k = 0; k < NHOT; ++k) {
								if (hv[k]) {
									const bool is = d == hk[k];
									const u64 m = __ballot(is);
									if (is)
										pos[r] = hcur[k] + mbcnt64(m);
									hcur[k] += (u32)__popcll(m);
									mine |= is;
								}
							}
						}
						if (!mine)
							pos[r] = __hip_atomic_fetch_add(&wc[d], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
					} else {
						pos[r] = __hip_atomic_fetch_add(&wc[d], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
					}
				}
			}
#pragma unroll
			for (int r = 0; r < SB; ++r) {
				const u32 o = wo + (r0 + r) * 64;
				if (full || o < cnt) {
					*(KT *)(sm.stage_raw + stage_swz<!HAS_VAL>(pos[r] * (u32)sizeof(KT))) = cur[r];
					if constexpr (HAS_VAL)
					{
						const u32 sh = 16 * ((r0 + r) & 1);
						if constexpr (RANK1)
							rk[(r0 + r) >> 1] = (rk[(r0 + r) >> 1] & ~(0xFFFFu << sh)) | (pos[r] << sh);
						else
							posp[(r0 + r) >> 1] |= pos[r] << sh;
					}
				}
			}
		};
		if constexpr (KEEP) {
#pragma unroll
			for (int r0 = 0; r0 < KPT; r0 += SB) {
				KT cur[SB];
#pragma unroll
				for (int r = 0; r < SB; ++r)
					cur[r] = keep[r0 + r];
				stage_batch(cur, r0);
			}
		} else {
			// two batches of loads are in flight while one is ranked and staged
			KT b0[SB], b1[SB];
			load_batch(b0, 0);
#pragma unroll
			for (int r0 = 0; r0 < KPT; r0 += 2 * SB) {
				load_batch(b1, r0 + SB);
				stage_batch(b0, r0);
				if (r0 + 2 * SB < KPT)
					load_batch(b0, r0 + 2 * SB);
				stage_batch(b1, r0 + SB);
			}
		}
		// Key + payload passes, whole tiles: the payloads are requested NOW, into the registers the keys have just left, so that
		// they cross the memory system while the keys are written out (requested behind the keys' write-out, a tile's 128 KiB
		// of payloads are 11 k cycles in which the workgroup does nothing else).
		// (A/B on one box, 2^28 f32 keys -> ranks: 3.29 against 3.35 ms on random bits, 3.51 against 3.72 and 3.53 against 3.69
		// on the two skewed inputs; keys + payload 4.51 against 4.57)
		constexpr bool PREV = HAS_VAL && KEEP && full;
		VT vpre[PREV ? KPT : 1];
		bool pre = false;
		if constexpr (PREV) {
			pre = !(flags & SCATTER_GEN_INDEX);
			if (pre) {
#pragma unroll
				for (int r = 0; r < KPT; ++r)
					vpre[r] = elem(vin, (SEG && HAS_VAL) ? base + vdiff : base, r, wo);
			}
		}
		__syncthreads();
		if (TL && tid == 0)
			tl[(u64)stile * 16 + 4 + 2 * t] = __builtin_readcyclecounter();

		// write out.  The staged tile is sorted by digit and consecutive staged elements of one digit go to
		// consecutive addresses: a lane takes CHUNK consecutive elements and, when they share a digit
		// (first == last), stores them with one wide store; chunks straddling a run boundary go element-wise.
		u32 pk[HAS_VAL ? KPT / CHUNK : 1];
#pragma unroll
		for (int j = 0; j < KPT / CHUNK; ++j) {
			if (j % 4 == 0)
				__builtin_amdgcn_sched_barrier(0);   // keep a few chunks' registers alive at a time
			const u32 i0 = opaque(CHUNK * tid) + CHUNK * j * BLOCK;   // (recomputed: kept across the tile, the indices cost registers)
			KT kv[CHUNK];
			u32 d[CHUNK];
			{
				typedef KT kvec_t __attribute__((ext_vector_type(CHUNK)));
				const kvec_t x = *(const kvec_t *)(sm.stage_raw + stage_swz<!HAS_VAL>(i0 * (u32)sizeof(KT)));
#pragma unroll
				for (int e = 0; e < CHUNK; ++e)
					kv[e] = x[e];
			}
#pragma unroll
			for (int e = 0; e < CHUNK; ++e)
				d[e] = digit2<DIG>(kv[e], ka, shift);
			if constexpr (HAS_VAL) {
				u32 p = 0;
#pragma unroll
				for (int e = 0; e < CHUNK; ++e)
					p |= d[e] << (8 * e);
				pk[j] = p;
			}
			if (!(flags & SCATTER_SKIP_KEYS) && !(TL && (flags & SCATTER_DBG_NOSTORE))) {
				const bool whole = full || i0 + CHUNK <= cnt;
				if constexpr (NARROW) {
					KTO ov[CHUNK];
#pragma unroll
					for (int e = 0; e < CHUNK; ++e)
						ov[e] = (KTO)(kdf_apply(kv[e], ka) >> oshift);
					if (sizeof(KTO) * CHUNK >= 4 && whole && d[0] == d[CHUNK - 1]) {
						store_chunk<KTO, CHUNK>(kout + (ST)(delta[d[0]] + i0), ov);
					} else {
#pragma unroll
						for (int e = 0; e < CHUNK; ++e)
							if (full || i0 + e < cnt)
								kout[(ST)(delta[d[e]] + i0 + e)] = ov[e];
					}
				} else if (sizeof(KT) >= 4 && whole && d[0] == d[CHUNK - 1]) {
					store_chunk<KT, CHUNK>(kout + (ST)(delta[d[0]] + i0), kv);
				} else {
#pragma unroll
					for (int e = 0; e < CHUNK; ++e)
						if (full || i0 + e < cnt)
							kout[(ST)(delta[d[e]] + i0 + e)] = kv[e];
				}
			}
		}
		if constexpr (HAS_VAL) {
			// payloads: same positions, through the same staging area
			const bool gen_index = (flags & SCATTER_GEN_INDEX) != 0;
			__syncthreads();
			if (TL && tid == 0)
				tl[(u64)stile * 16 + 8] = __builtin_readcyclecounter();   // keys written out
#pragma unroll
			for (int r0 = 0; r0 < KPT; r0 += SB) {
				VT val[SB];
#pragma unroll
				for (int r = 0; r < SB; ++r) {
					const u32 o = wo + (r0 + r) * 64;
					if (PREV && pre)
						val[r] = vpre[PREV ? r0 + r : 0];
					else
						val[r] = gen_index ? (VT)(base + o) : ((full || o < cnt) ? elem(vin, (SEG && HAS_VAL) ? base + vdiff : base, r0 + r, wo) : (VT)0);
				}
#pragma unroll
				for (int r = 0; r < SB; ++r) {
					const u32 o = wo + (r0 + r) * 64;
					if (full || o < cnt)
					{
						const u32 packed = RANK1 ? rk[RANK1 ? (r0 + r) >> 1 : 0] : posp[RANK1 ? 0 : (r0 + r) >> 1];
						*(VT *)(sm.stage_raw + stage_swz<!HAS_VAL>(((packed >> (16 * ((r0 + r) & 1))) & 0xFFFFu) * (u32)sizeof(VT))) = val[r];
					}
				}
			}
			__syncthreads();
			if (TL && tid == 0)
				tl[(u64)stile * 16 + 9] = __builtin_readcyclecounter();   // payloads staged
#pragma unroll
			for (int j = 0; j < KPT / CHUNK; ++j) {
				if (j % 4 == 0)
					__builtin_amdgcn_sched_barrier(0);
				const u32 i0 = opaque(CHUNK * tid) + CHUNK * j * BLOCK;   // (recomputed: kept across the tile, the indices cost registers)
				VT vv[CHUNK];
				{
					typedef VT vvec_t __attribute__((ext_vector_type(CHUNK)));
					const vvec_t x = *(const vvec_t *)(sm.stage_raw + stage_swz<!HAS_VAL>(i0 * (u32)sizeof(VT)));
#pragma unroll
					for (int e = 0; e < CHUNK; ++e)
						vv[e] = x[e];
				}
				const u32 d0 = pk[j] & 0xFFu, dl = (pk[j] >> (8 * (CHUNK - 1))) & 0xFFu;
				const bool whole = full || i0 + CHUNK <= cnt;
				if constexpr (SEG) {
					// (the payloads' slot arrays: SegArgs::v_off_*; zero wherever the slots lie as the keys' do)
					if (whole && d0 == dl) {
						store_chunk<VT, CHUNK>(vout + (ST)(delta[d0] + i0 + (d0 < vadj_n ? vadj_lo : vadj_hi)), vv);
					} else {
#pragma unroll
						for (int e = 0; e < CHUNK; ++e)
							if (full || i0 + e < cnt) {
								const u32 de = (pk[j] >> (8 * e)) & 0xFFu;
								vout[(ST)(delta[de] + i0 + e + (de < vadj_n ? vadj_lo : vadj_hi))] = vv[e];
							}
					}
				} else if (whole && d0 == dl) {
					store_chunk<VT, CHUNK>(vout + (ST)(delta[d0] + i0), vv);
				} else {
#pragma unroll
					for (int e = 0; e < CHUNK; ++e)
						if (full || i0 + e < cnt)
							vout[(ST)(delta[(pk[j] >> (8 * e)) & 0xFFu] + i0 + e)] = vv[e];
				}
			}
		}
		__syncthreads();   // staging reads done before the next tile is staged
		if (TL && tid == 0)
			tl[(u64)stile * 16 + 5 + 2 * t] = __builtin_readcyclecounter();
	};
#pragma unroll
	for (int t = 0; t < TPS; ++t) {
		const u64 base = beg + (u64)t * C::TILE;
		if (!(t < (int)tps && base < end))
			break;
		const u32 cnt = (end - base) < (u64)C::TILE ? (u32)(end - base) : (u32)C::TILE;
		if (cnt == (u32)C::TILE)
			do_tile(std::true_type{}, t, base, cnt);
		else
			do_tile(std::false_type{}, t, base, cnt);
	}
}

}  // namespace rsx
