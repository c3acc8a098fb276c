// rsx_pass2w.hpp -- the LEVEL-2 pass of 8-byte keys in whole 64-byte atoms (round 6), gfx950: one body, two routes.
//
// What a level-2 pass has to do (rsx_hybrid.hpp): level 1 has left buckets; every key of bucket b goes to slot
// (b, d), d = eight bits of its derived key at the bucket's shift, as the value the leaves sort -- the low word of the derived key
// (four bytes per key: what is undecided below the two digits fits 32 bits) or the whole element image.  The reference makes the same
// trip per kept column (radix_sort.hpp:82-90); the order INSIDE a slot is free (the leaves sort by value, equal keys are equal bits).
//
// rsx_pass16a_kernel's scheme (rsx_pass16.hpp) for 8-byte keys: a workgroup owns a contiguous range of the tile table and CARRIES,
// per digit, what does not fill a 64-byte atom into its next tile; a digit's place in its slot is one returning global atomic per
// (tile, digit) on the slot's front cursor; what is still carried when the bucket changes or the range ends goes to the slot's back
// (its own cursor); every store of the pass is a whole, aligned atom, and the leaves read a slot from both ends.  One LDS atomic
// per key: the counting atomic's return value is the key's rank in its digit's run (kept, sixteen bits per key).  Two workgroups
// of 1024 threads per CU (one tile's loads overlap the other's ranking and stores).
//
// Through round 5 this pass was an instantiation of rsx_scatter2_kernel<u64, ..., KTO, SEG> (chained tiles, ragged runs): 0.806 ms
// for 2^28 keys into four-byte slots = 0.50 of the HBM peak, traffic 1.07 x.
//
// The two routes differ in where tiles, shifts, slots and verdicts live; `Policy` says:
//   bool     go() const                          nothing speaks against running
//   u32      ntiles() const, per(u32 grid) const tiles in all, tiles per workgroup
//   Tile     tile(u32 t) const                   { const KT *keys; u32 cnt; u32 bucket; }
//   u32      shift(u32 bucket) const             the level-2 digit's bit position
//   u32      cap(u32 bucket) const               values a slot of the bucket holds (its last `BACK` places are its back)
//   u32      slot(u32 bucket, u32 d) const       first place of slot (bucket, d) in the output array
//   u32     *front(u32 bucket, u32 d) const, *back(u32 bucket, u32 d) const      the slot's two cursors
//   void     lost(u32 what) const                the attempt is lost (a slot overflowed: 1 its front, 2 its back)
//   u32      dump() const                        where the runs of a lost attempt go (a tile of padding behind the slots)
#pragma once

#include "rsx_kernels.hpp"
#include "rsx_scatter2.hpp"

namespace rsx {

template <typename KT> struct Pass2wTile {
	const KT *keys;
	u32 cnt, bucket;
};

// OT = u32: the low word of the derived key, atoms of sixteen values, twelve keys per lane (48 KiB staged);
// OT = u64: the element image, atoms of eight keys, six keys per lane (the same 48 KiB)
// KPT_ = 24 (four-byte values in AND out, rsx_pass64a_kernel<u32, u32>): 24 Ki-key tiles, 96 KiB staged, ONE workgroup per CU as
// rsx_pass32a_kernel has it -- the per-tile costs (barriers, 256 global atomics) then stand against twice the bytes
template <typename OT, int KPT_ = (sizeof(OT) == 4 ? 12 : 6)> struct Pass2wCfg {
	static constexpr int BLOCK = 1024, KPT = KPT_, TILE = BLOCK * KPT, SB = sizeof(OT) == 4 ? 4 : 3;
	static constexpr u32 ATOM = 64 / sizeof(OT), VEC = 16 / sizeof(OT);
	static constexpr int STAGE = TILE + 256 * 2 * ((int)VEC - 1);   // + what the 16-byte alignment of 256 runs can cost
	static constexpr int WGS = STAGE * (int)sizeof(OT) <= 56 * 1024 ? 2 : 1;   // workgroups per CU
	static constexpr int GRID = 256 * WGS;
	static constexpr u32 BACK = 128;   // places at a slot's end for what is carried when a range ends: a bucket's tiles are shared by at most eight workgroups
};

template <typename OT, typename C_ = Pass2wCfg<OT> > struct Pass2wSmem {
	typedef C_ C;
	__attribute__((aligned(16))) OT stage[C::STAGE];
	__attribute__((aligned(16))) OT carry[256][C::ATOM];
	u32 cell[2][256];   // per digit: count, then the run's tile-local start; tiles alternate between the two
	u32 delta[256];     // slot position of a body value minus its tile-local position
	u32 info[256];      // carried before (5 bits) | head (5) | tail (5) | atom completed | enough for an atom | offset in the region
	unsigned short rbeg[256], bbeg[256], bend[256];
	unsigned char group_digit[C::STAGE / C::VEC];
	u32 wsum[4];
};

template <typename KT, typename OT, bool NT_LOADS, typename Policy, typename C_ = Pass2wCfg<OT> >
__device__ __forceinline__ void pass2w_body(const Policy pol, OT *__restrict__ kout, const KdfArgs<KT> ka, Pass2wSmem<OT, C_> &sm)
{
	// (KT = u32, OT = u32: the level-1 slots hold the low words of the derived keys already, SegCtl::narrow == 2 -- identity KDF)
	static_assert((sizeof(KT) == 8 && (sizeof(OT) == 4 || sizeof(OT) == 8)) || (sizeof(KT) == 4 && sizeof(OT) == 4),
	              "8-byte keys into four- or eight-byte slots, or four-byte values into four-byte slots");
	constexpr int VIN = 16 / (int)sizeof(KT);
	typedef C_ C;
	constexpr int BLOCK = C::BLOCK, KPT = C::KPT, TILE = C::TILE, SB = C::SB;
	constexpr u32 VEC = C::VEC, ATOM = C::ATOM, BACK = C::BACK;
	static_assert(KPT % VIN == 0 && KPT % SB == 0, "whole 16-byte loads, whole staging batches");
	if (!pol.go())
		return;
	const u32 ntiles = pol.ntiles(), per = pol.per(gridDim.x);
	const u32 t0 = blockIdx.x * per, t1 = t0 + per < ntiles ? t0 + per : ntiles;
	if (t0 >= t1)
		return;
	const u32 tid0 = threadIdx.x;
	auto sidx = [](u32 pos) { return stage_swz<true>(pos * (u32)sizeof(OT)); };
	auto staged = [&](u32 pos) -> OT & { return *(OT *)((char *)sm.stage + sidx(pos)); };
	u32 cc = 0;   // digit thread: values of its digit carried from the tiles before
	u32 bucket = pol.tile(t0).bucket;
	if (tid0 < 256)
		sm.cell[0][tid0] = 0;
	// what is carried goes to the back of its slot (the range ends, or the next tile lies in another bucket)
	auto flush = [&]() {
		const u32 tid = tid0, cd = tid >> 2, part = tid & 3u;
		__syncthreads();
		if (tid < 256) {
			u32 inf = 0, dest = 0;
			if (cc) {
				const u32 cap = pol.cap(bucket);
				const u32 pos = __hip_atomic_fetch_add(pol.back(bucket, tid), cc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				if (pos + cc > BACK)
					pol.lost(2u);
				else {
					inf = cc;
					dest = pol.slot(bucket, tid) + (cap - BACK) + pos;
				}
			}
			sm.info[tid] = inf;
			sm.delta[tid] = dest;
			cc = 0;
		}
		__syncthreads();
		{
			const u32 nk = sm.info[cd], dest = sm.delta[cd];
#pragma unroll
			for (u32 e = 0; e < VEC; ++e) {
				const u32 k = part * VEC + e;
				if (k < nk)
					kout[dest + k] = sm.carry[cd][k];
			}
		}
		__syncthreads();
	};
	__syncthreads();
	// (the two workgroups of a CU start half a tile apart: started together they would load, rank and store in step)
	if (C::WGS == 2 && blockIdx.x >= gridDim.x / 2)
		__builtin_amdgcn_s_sleep(127);
	KT keep[KPT];
	for (u32 t = t0; t < t1; ++t) {
		// (everything a tile derives from the thread index is derived from an opaque copy of it, made per tile: as loop invariants
		// the LDS addresses of a dozen tables would be hoisted in front of the loop and spilled there)
		u32 tid = tid0;
		asm volatile("" : "+v"(tid));
		const u32 lane = tid & 63, wid = tid >> 6;
		const u32 cd = tid >> 2, part = tid & 3u;   // the copying threads: digit, quarter of an atom
		u32 *const cell = sm.cell[(t - t0) & 1u];
		const Pass2wTile<KT> st = pol.tile(t);
		if (st.bucket != bucket) {
			flush();
			bucket = st.bucket;
		}
		const u32 cnt = st.cnt;
		const bool full = cnt == (u32)TILE && (((uintptr_t)st.keys) & 15) == 0;
		const u32 shift = pol.shift(bucket);
		if (full) {
			typedef KT vec_t __attribute__((ext_vector_type(VIN)));
			const vec_t *vp = (const vec_t *)st.keys + tid;
#pragma unroll
			for (int i = 0; i < KPT / VIN; ++i) {
				const vec_t v = NT_LOADS ? __builtin_nontemporal_load(&vp[i * BLOCK]) : vp[i * BLOCK];
#pragma unroll
				for (int e = 0; e < VIN; ++e)
					keep[VIN * i + e] = v[e];
			}
		} else {
#pragma unroll
			for (int r = 0; r < KPT; ++r) {
				const u32 o = tid + r * BLOCK;
				keep[r] = o < cnt ? st.keys[o] : (KT)0;
			}
		}
		// (four-byte slots: derived once, the digit and the value are bit fields of the derived key from here on; eight-byte slots
		// take the element image as it is and derive for the digit -- three instructions, twice)
		if constexpr (sizeof(OT) == 4) {
#pragma unroll
			for (int r = 0; r < KPT; ++r)
				keep[r] = kdf_apply(keep[r], ka);
		}
		auto digit_of_key = [&](const KT k, const u32 sh) -> u32 {
			return (u32)((sizeof(OT) == 4 ? k : kdf_apply(k, ka)) >> sh) & 0xFFu;
		};
		u32 rk[KPT / 2];
		auto count = [&](auto full_c) {
			constexpr bool FULL = decltype(full_c)::value;
#pragma unroll
			for (int r = 0; r < KPT; ++r) {
				u32 mine = 0;
				if (FULL || tid + r * BLOCK < cnt)
					mine = __hip_atomic_fetch_add(&cell[digit_of_key(keep[r], shift)], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
				rk[r >> 1] = (r & 1) ? rk[r >> 1] | (mine << 16) : mine;
			}
		};
		if (full)
			count(std::true_type{});
		else
			count(std::false_type{});
		__syncthreads();

		// ---- digit thread d: what of (carried + this tile's) values goes out, where in the slot, where in the staging area
		u32 base = 0;
		{
			u32 rlen = 0, rstart = 0;
			if (tid < 256) {
				const u32 c = cell[tid];
				u32 h, body = 0, tail = 0, atom = 0;
				const bool enough = cc + c >= ATOM;
				if (enough) {
					h = cc ? ATOM - cc : 0u;      // the head completes the carried atom
					atom = cc ? 1u : 0u;
					body = (c - h) & ~(ATOM - 1u);
					tail = (c - h) & (ATOM - 1u);
				} else {
					h = c;                        // too few for an atom: all of it joins the carried values
				}
				const u32 mm = atom * ATOM + body;
				if (mm)   // (issued first: it crosses the fabric while the layout is made)
					base = __hip_atomic_fetch_add(pol.front(bucket, tid), mm, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				const u32 o = (VEC - (h & (VEC - 1u))) & (VEC - 1u);   // the run starts `o` values into its region: the body then starts on a 16-byte boundary
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
				cell[tid] = rb;
				sm.cell[((t - t0) & 1u) ^ 1u][tid] = 0;   // (the next tile's counters)
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
			const u32 cap = pol.cap(bucket);
			const u32 atom = (sm.info[tid] >> 15) & 1u, bb = sm.bbeg[tid], mm = atom * ATOM + (sm.bend[tid] - bb);
			u32 dest = pol.slot(bucket, tid) + base + atom * ATOM;   // of the body's first value
			if (mm && base + mm > cap - BACK) {
				pol.lost(1u);   // the slot is too small: the attempt is lost, its values go to the dump area behind the slots
				dest = pol.dump() + ATOM;
			}
			sm.delta[tid] = dest - bb;
		}
		u32 shift_b = shift;
		asm volatile("" : "+s"(shift_b));   // (the digits are computed again, not kept across the barriers: 64 registers per lane)
		auto stage_keys = [&](auto full_c) {
			constexpr bool FULL = decltype(full_c)::value;
#pragma unroll
			for (int r0 = 0; r0 < KPT; r0 += SB) {
				u32 pos[SB];
#pragma unroll
				for (int r = 0; r < SB; ++r) {
					pos[r] = 0;
					if (FULL || tid + (r0 + r) * BLOCK < cnt)
						pos[r] = cell[digit_of_key(keep[r0 + r], shift_b)] + ((rk[(r0 + r) >> 1] >> (16 * ((r0 + r) & 1))) & 0xFFFFu);
				}
#pragma unroll
				for (int r = 0; r < SB; ++r) {
					if (FULL || tid + (r0 + r) * BLOCK < cnt)
						staged(pos[r]) = (OT)keep[r0 + r];
				}
			}
		};
		if (full)
			stage_keys(std::true_type{});
		else
			stage_keys(std::false_type{});
		__syncthreads();

		// ---- out: the completed atoms (a quarter per copying thread: carried values, then the head of the run) ...
		{
			const u32 inf = sm.info[cd];
			const u32 ccd = inf & 31u, atomd = (inf >> 15) & 1u;
			if (atomd) {
				const u32 rb = sm.rbeg[cd];
				typedef OT ovec_t __attribute__((ext_vector_type(VEC)));
				typedef ovec_t avec_t __attribute__((aligned(16)));
				ovec_t w;
#pragma unroll
				for (u32 e = 0; e < VEC; ++e) {
					const u32 k = part * VEC + e;
					w[e] = k < ccd ? sm.carry[cd][k] : staged(rb + (k - ccd));
				}
				*(avec_t *)(kout + (u32)(sm.delta[cd] + sm.bbeg[cd] - ATOM + part * VEC)) = w;
			}
		}
		// ... and the bodies: every group of VEC staged values that lies in one is a quarter of an aligned atom
		{
			const u32 total = (u32)__builtin_amdgcn_readfirstlane((int)sm.wsum[0]) + sm.wsum[1] + sm.wsum[2] + sm.wsum[3];
#pragma unroll 1
			for (u32 i0 = VEC * tid; i0 < total; i0 += VEC * BLOCK) {
				const u32 d = sm.group_digit[i0 / VEC];
				if (i0 >= sm.bbeg[d] && i0 < sm.bend[d]) {
					typedef OT ovec_t __attribute__((ext_vector_type(VEC)));
					typedef ovec_t avec_t __attribute__((aligned(16)));
					*(avec_t *)(kout + (u32)(sm.delta[d] + i0)) = *(const ovec_t *)((const char *)sm.stage + sidx(i0));
				}
			}
		}
		// ---- what stays: the tail of the run (or, with too few values for an atom, all of the run behind what was carried).  (No
		// barrier in front: carry[d][VEC part ..] was read for the atom above by this very thread.  None behind: the next tile stages
		// -- and reads the carried values -- behind three barriers of its own.)
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
	flush();
}

}  // namespace rsx
