// rsx_logroute.hpp -- two MSB passes and leaves for 8-byte keys whose BYTES do not spread but whose MAGNITUDES do (round 6), gfx950.
//
// The reference's loop makes one trip through memory per kept column whatever the keys look like (radix_sort.hpp:82-90).  The
// two-level routes of rsx_hybrid.hpp make two trips and a leaf pass instead, but they cut the keys by BYTES, and keys drawn
// from a heavy-tailed distribution -- BASELINE.json's cfg 3 (iv): Zipf-like u64 keys, log-uniform over [1, 2^40) -- put a
// thirty-second of the array into ONE value of the top kept byte's bucket 0 and almost nothing into the others: no slot scheme
// by bytes takes them, and through round 5 they went one pass per kept column (88 B/key, 5.3 ms for 2^28 keys).
//
// What spreads such keys is an order-preserving digit of (bit length, leading mantissa bits) -- a float's exponent and top
// fraction bits, made from an integer with one count-leading-zeros:
//
//   small keys   derived keys below 2^14 (after the constant top bits are dropped) are COUNTED, in a 16384-entry table per
//                workgroup of the histogram kernel; the sorted array's beginning is that table written out (rsx_log_fill_kernel:
//                the precedent is rsx_fill_runs_kernel for one-column keys) -- they are read once and written once;
//   level 1      the other keys go by  d1 = (bit length - 15) << m | the m bits below the leading one  (m = 3 for keys below
//                2^44: 208 buckets for 2^40) into buckets whose EXACT sizes the histogram kernel counted: no slots, no slack;
//                whole 64-byte atoms from the bucket's front, what a workgroup still carries at its range's end from the
//                bucket's back (the two meet exactly);
//   level 2      inside a level-1 bucket by the next eight bits below those (a shift per bucket), into slots with slack, as
//                FOUR-byte values -- what is left undecided is at most 32 bits; whole atoms of sixteen values, two cursors per slot;
//   leaves       a slot's values are placed by their top twelve undecided bits and finished by the register networks of
//                rsx_leaf16.hpp (slots with at most twelve undecided bits are done by the placement alone), the upper bits
//                come back from the slot's digits, the element images through kdf_invert.
//
// The same stable order as the reference's P passes: equal derived keys of a keys-only sort are equal bit patterns.  Which
// buffer the result lies in follows the reference's parity rule (radix_sort.hpp:92) from the kept BYTE columns, which the
// histogram kernel finds exactly (OR over all keys of key ^ first key: a column is kept iff some key differs from the first
// in it, radix_sort.hpp:64-70), as it finds the pre-sorted exit (:60-62: no descent among neighbours).
//
// Everything is device-scheduled behind a sample (one workgroup) that says whether the route is worth trying; the plan kernel
// checks every capacity exactly and a slot that overflows in level 2 sets LogCtl::fail -- the caller's first buffer is only
// read until then, so the ordinary histogram-first sort starts from untouched input.
#pragma once

#include "rsx_kernels.hpp"
#include "rsx_hybrid.hpp"
#include "rsx_scatter2.hpp"
#include "rsx_leaf16.hpp"
#include "rsx_pass2w.hpp"

namespace rsx {

#ifndef RSX_LOG_C
#define RSX_LOG_C 14
#endif
constexpr u32 LOG_C = RSX_LOG_C;           // derived keys below 2^LOG_C are counted, not moved (2^28 Zipf-like keys, tools/ubench/log_probe:
                                           // 2^12 2.39-2.41 ms, 2^13 2.34-2.37, 2^14 2.33-2.35 -- 64 KiB of counters per workgroup of the histogram kernel)
static_assert(LOG_C >= 12 && LOG_C <= 14, "the level-2 digit lies below the level-1 digit's bits (LOG_C >= m + 8); the table fits the LDS");
constexpr u32 LOG_NSMALL = 1u << LOG_C;
constexpr u32 LOG_LEAF_CAP = 5120;         // values a leaf holds (rsx_log_leaf_kernel, LogLeafCfg<256, 5120, 12>) ...
constexpr u32 LOG_LEAF_CAP_BIG = 10240;    // ... and the shape for arrays beyond 2^28 + 2^24 keys (LogLeafCfg<512, 10240, 13>)
constexpr u32 LOG_BACK2 = 128;             // places at the end of every level-2 slot for what is carried when a range ends
constexpr u32 LOG_MAX_B = 44;              // keys that vary in more low bits than this do not leave 32 undecided bits or fewer

struct LogCtl {   // 256 bytes; zeroed by the host, then: sample -> histogram -> plan -> passes
	u32 go;        // sample: 1 the route is worth trying
	u32 B;         // sample: the derived keys differ from the first one in their low B bits only (checked on every key)
	u32 m;         // sample: mantissa bits of the level-1 digit
	u32 ndig;      // sample: level-1 digits in use ((B - LOG_C) << m)
	u32 fail;      // any kernel: the attempt is lost (a capacity, a key outside the low B bits, a slot that overflowed)
	u32 desc_cnt;  // histogram: descents among neighbours (0: pre-sorted, radix_sort.hpp:60-62)
	u32 or_lo, or_hi;       // histogram: OR over all keys of (derived key ^ first derived key)
	u32 key0_lo, key0_hi;   // sample: the first derived key
	u32 nsmall;    // plan: keys below 2^LOG_C
	u32 ntiles2;   // plan: tiles of the level-2 pass
	u32 ok;        // plan: 1 the passes may run
	u32 sorted;    // plan: the input is sorted (nothing runs, aux stays untouched)
	u32 per2;      // plan: tiles per workgroup of the level-2 pass
	u32 maxh1;     // plan: the largest level-1 bucket
	u32 ncols;     // plan: kept byte columns (radix_sort.hpp:64-70)
	u32 leaf_cap;  // host: values the leaf shape launched behind this attempt holds (LOG_LEAF_CAP or LOG_LEAF_CAP_BIG)
	u32 pad[46];
};
static_assert(sizeof(LogCtl) == 256, "LogCtl");

// [zeroed per sort: cnt, cur1 | written by the plan kernel: the rest]
struct LogTabs {
	u32 cnt[LOG_NSMALL + 256];        // histogram: the small keys' table, then the level-1 digits' counts
	u32 cur1[512];                    // level-1 pass: front [256] and back [256] cursors of the buckets
	u32 offs_small[LOG_NSMALL + 4];   // plan: exclusive scan of the small table (+ the total)
	u32 out1[256];                    // plan: where level-1 bucket d begins in the sorted array
	u32 reg1[256];                    // plan: where it begins in the level-1 array (a multiple of eight keys)
	u32 cap2[256];                    // plan: capacity of each of its 256 level-2 slots
	u32 base2[256];                   // plan: where its first level-2 slot begins in the slot array
	u32 tb2[260];                     // plan: its first level-2 tile (+ the total)
};
struct LogTile {
	u32 beg, cnt, bucket, pad;
};

// the level-1 digit of a derived key cut to its low B bits, kk >= 2^LOG_C
__device__ __forceinline__ u32 log_digit(u64 kk, u32 m)
{
	const u32 b = 64u - (u32)__builtin_clzll(kk);   // bit length, > LOG_C
	return ((b - LOG_C - 1u) << m) | ((u32)(kk >> (b - 1u - m)) & ((1u << m) - 1u));
}
// ... its bit length, and the shift of the level-2 digit of bucket d (the eight bits below the level-1 digit's)
__device__ __forceinline__ u32 log_blen(u32 d, u32 m) { return LOG_C + 1u + (d >> m); }
__device__ __forceinline__ u32 log_shift2(u32 d, u32 m) { return log_blen(d, m) - 1u - m - 8u; }   // (>= 0: m <= 4, bit length >= 13)
// capacity of a level-2 slot for buckets of `mean` values: an eighth above it, or seven standard deviations, + the back
__device__ __forceinline__ u32 log_cap2(u32 mean)
{
	const u32 r = (u32)sqrtf((float)mean) + 1u;
	const u32 slack = mean / 8u > 7u * r + 8u ? mean / 8u : 7u * r + 8u;
	return (mean + slack + LOG_BACK2 + 31u) & ~31u;
}

// ---- the sample ---------------------------------------------------------------------------------------------------------------
// One workgroup, 64 places of 128 consecutive keys (as rsx_blind_precheck_kernel).  go = the sampled derived keys agree above
// bit B (25 <= B <= 44), and no level-1 digit holds more of the sample than a leaf-sized share allows.
template <typename KT>
__global__ __launch_bounds__(1024) void rsx_log_sample_kernel(const KT *__restrict__ src, u64 n, KdfArgs<KT> ka, LogCtl *__restrict__ ctl,
                                                              u32 leaf_cap = LOG_LEAF_CAP)
{
	static_assert(sizeof(KT) == 8, "8-byte keys");
	constexpr u32 S = 8, NS = 1024 * S;
	__shared__ u32 hs[256];
	__shared__ u32 s_or[2], s_max, s_small;
	const u32 tid = threadIdx.x;
	if (tid < 256)
		hs[tid] = 0;
	if (tid == 0)
		s_or[0] = s_or[1] = s_max = s_small = 0;
	__syncthreads();
	const u64 i0 = ((n - 16 * S) / 63) * (tid >> 4) + (tid & 15u) * S;
	const u64 key0 = (u64)kdf_apply(src[0], ka);
	u64 k[S], v = 0;
#pragma unroll
	for (u32 e = 0; e < S; ++e) {
		k[e] = (u64)kdf_apply(src[i0 + e], ka);
		v |= k[e] ^ key0;
	}
#pragma unroll
	for (int off = 32; off > 0; off >>= 1)
		v |= ((u64)(u32)__shfl_xor((int)(u32)(v >> 32), off) << 32) | (u32)__shfl_xor((int)(u32)v, off);
	if ((tid & 63) == 0) {
		atomicOr(&s_or[0], (u32)v);
		atomicOr(&s_or[1], (u32)(v >> 32));
	}
	__syncthreads();
	const u64 vary = ((u64)s_or[1] << 32) | s_or[0];
	const u32 B = vary ? 64u - (u32)__builtin_clzll(vary) : 0u;
	u32 m = 0;
	bool go = B >= 25u && B <= LOG_MAX_B;
	if (go) {
		const u32 nexp = B - LOG_C;
		while (m < 4u && (nexp << (m + 1u)) <= 256u)
			++m;
		go = B <= 41u + m;   // (at most 32 bits below the two digits)
	}
	if (go) {
		const u64 lowmask = ((u64)1 << B) - 1u;
		u32 small = 0;
#pragma unroll
		for (u32 e = 0; e < S; ++e) {
			const u64 kk = k[e] & lowmask;
			if (kk < LOG_NSMALL)
				++small;
			else
				atomicAdd(&hs[log_digit(kk, m)], 1u);
		}
		if (small)
			atomicAdd(&s_small, small);
	}
	__syncthreads();
	if (go && tid < 256) {
		u32 mx = hs[tid];
#pragma unroll
		for (int off = 32; off > 0; off >>= 1) {
			const u32 y = (u32)__shfl_xor((int)mx, off);
			mx = y > mx ? y : mx;
		}
		if ((tid & 63) == 0)
			atomicMax(&s_max, mx);
	}
	__syncthreads();
	if (tid == 0) {
		if (go) {
			// the largest level-1 bucket may hold 256 leaves' worth of keys: its share of the sample, a quarter and four standard
			// deviations on top (the plan kernel decides exactly; this only keeps hopeless inputs from paying for the histogram)
			const float share = (float)NS * 256.0f * (float)leaf_cap / (float)n;
			const float lim = share + 0.25f * share + 4.0f * sqrtf(share) + 8.0f;
			go = (float)s_max <= lim;
		}
		ctl->go = go ? 1u : 0u;
		ctl->B = B;
		ctl->m = m;
		ctl->ndig = go ? (B - LOG_C) << m : 0u;
		ctl->key0_lo = (u32)key0;
		ctl->key0_hi = (u32)(key0 >> 32);
		ctl->leaf_cap = leaf_cap;
	}
}

// ---- the histogram ------------------------------------------------------------------------------------------------------------
// One read of the keys (radix_sort.hpp:47-58): the small keys' table and the level-1 digits' counts (one LDS atomic per key), the
// descents among neighbours, the OR of every key's difference from the first.  src must be 16-byte aligned.
template <typename KT>
__global__ __launch_bounds__(1024, 2) void rsx_log_hist_kernel(const KT *__restrict__ src, u64 n, KdfArgs<KT> ka,
                                                              LogCtl *__restrict__ ctl, LogTabs *__restrict__ tabs)
{
	static_assert(sizeof(KT) == 8, "8-byte keys");
	if (ctl->go != 1u)
		return;
	constexpr u32 NB = LOG_NSMALL + 256;
	__shared__ u32 cnt[NB];
	__shared__ u32 s_desc, s_or[2];
	const u32 tid = threadIdx.x, lane = tid & 63;
	for (u32 i = tid; i < NB; i += 1024)
		cnt[i] = 0;
	if (tid == 0)
		s_desc = s_or[0] = s_or[1] = 0;
	__syncthreads();
	const u32 B = ctl->B, m = ctl->m;
	const u64 key0 = ((u64)ctl->key0_hi << 32) | ctl->key0_lo;
	const u64 lowmask = ((u64)1 << B) - 1u;
	u64 vor = 0;
	u32 desc = 0;
	auto count = [&](const u64 k) {
		vor |= k ^ key0;
		const u64 kk = k & lowmask;
		const u64 kd = kk < LOG_NSMALL ? (u64)LOG_NSMALL : kk;   // (a defined digit for the lanes that take the other arm)
		const u32 idx = kk < LOG_NSMALL ? (u32)kk : LOG_NSMALL + (log_digit(kd, m) & 255u);
		atomicAdd(&cnt[idx], 1u);
	};
	typedef u64 vec_t __attribute__((ext_vector_type(2)));
	const vec_t *vp = (const vec_t *)src;
	const u64 nvec = n >> 1;
	// (wave-uniform trips: every lane of a wave is in the loop while the wave's first pair exists, so the DPP shift below always
	// reads a live lane)
	for (u64 base = (u64)blockIdx.x * 1024 + (tid & ~63u); base < nvec; base += (u64)gridDim.x * 1024) {
		const u64 i = base + lane;
		const bool act = i < nvec;
		vec_t x = {0, 0};
		if (act)
			x = vp[i];
		// the key behind this lane's pair: the next lane's first (a DPP wave shift); the wave's last lane, and the lane in front
		// of an odd array's last key, load it
		const bool has_next = act && 2 * i + 2 < n;
		const bool own = has_next && (lane == 63 || i + 1 >= nvec);
		u64 edge = 0;
		if (own)
			edge = (u64)src[2 * i + 2];
		const u64 k0 = (u64)kdf_apply((KT)x[0], ka), k1 = (u64)kdf_apply((KT)x[1], ka);
		u64 nx = ((u64)wave_next_lane((u32)(k0 >> 32)) << 32) | wave_next_lane((u32)k0);
		if (own)
			nx = (u64)kdf_apply((KT)edge, ka);
		if (act) {
			desc += k1 < k0 ? 1u : 0u;
			desc += has_next && nx < k1 ? 1u : 0u;
			count(k0);
			count(k1);
		}
	}
	if ((n & 1) && blockIdx.x == 0 && tid == 0)
		count((u64)kdf_apply(src[n - 1], ka));
	// the workgroup's results
#pragma unroll
	for (int off = 32; off > 0; off >>= 1) {
		vor |= ((u64)(u32)__shfl_xor((int)(u32)(vor >> 32), off) << 32) | (u32)__shfl_xor((int)(u32)vor, off);
		desc += (u32)__shfl_xor((int)desc, off);
	}
	if (lane == 0) {
		if (desc)
			atomicAdd(&s_desc, desc);
		atomicOr(&s_or[0], (u32)vor);
		atomicOr(&s_or[1], (u32)(vor >> 32));
	}
	__syncthreads();
	for (u32 i = tid; i < NB; i += 1024) {
		const u32 c = cnt[i];
		if (c)
			atomicAdd(&tabs->cnt[i], c);
	}
	if (tid == 0) {
		if (s_desc)
			atomicAdd(&ctl->desc_cnt, s_desc > 0x7FFFFFFFu ? 0x7FFFFFFFu : s_desc);
		if (s_or[0])
			atomicOr(&ctl->or_lo, s_or[0]);
		if (s_or[1])
			atomicOr(&ctl->or_hi, s_or[1]);
	}
}

// ---- the plan -----------------------------------------------------------------------------------------------------------------
// One workgroup: the pre-sorted exit and the kept byte columns (radix_sort.hpp:60-70) into the plan (device and host copies),
// the scans (:72-80), every capacity, the level-2 tiles.
__global__ __launch_bounds__(1024) void rsx_log_plan_kernel(LogCtl *__restrict__ ctl, LogTabs *__restrict__ tabs,
                                                           LogTile *__restrict__ tiles, u64 n, u32 l1_cap, u32 l2_cap,
                                                           u32 tiles_cap, u32 tile2, u32 grid2, Plan *__restrict__ plan,
                                                           Plan *host_plan)
{
	if (ctl->go != 1u)
		return;
	__shared__ u32 s_w[16], s_h1[256], s_fail;
	const u32 tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
	const u64 vary = ((u64)ctl->or_hi << 32) | ctl->or_lo;
	const u32 B = ctl->B, m = ctl->m, ndig = ctl->ndig;
	const bool sorted = ctl->desc_cnt == 0;
	const bool inside = B >= 64u || (vary >> B) == 0;
	if (tid == 0) {
		s_fail = inside ? 0u : 1u;
		u32 nc = 0, cols[8] = {0, 0, 0, 0, 0, 0, 0, 0};
		for (u32 c = 0; c < 8; ++c)
			if ((vary >> (8 * c)) & 0xFFu)
				cols[nc++] = c;
		ctl->ncols = nc;
		ctl->sorted = sorted ? 1u : 0u;
		if (inside) {
			Plan *const out[2] = {plan, host_plan};
			for (int j = 0; j < 2; ++j) {
				if (!out[j])
					continue;
				out[j]->ncols = nc;
				out[j]->sorted = sorted ? 1u : 0u;
				for (u32 i = 0; i < 8; ++i)
					out[j]->cols[i] = cols[i];
				out[j]->hot = 0;
				out[j]->vary_lo = (u32)vary;
				out[j]->vary_hi = (u32)(vary >> 32);
				out[j]->hyb = HYB_NONE;
				out[j]->max1 = 0;
			}
		}
	}
	__syncthreads();
	if (sorted || !inside) {
		if (tid == 0 && !inside)
			ctl->fail = 1u;
		return;
	}
	// exclusive scan over 1024 threads
	auto scan1024 = [&](const u32 v, u32 &total) -> u32 {
		u32 x = v;
#pragma unroll
		for (int off = 1; off < 64; off <<= 1) {
			const u32 y = __shfl_up(x, off);
			if (lane >= (u32)off)
				x += y;
		}
		__syncthreads();
		if (lane == 63)
			s_w[wid] = x;
		__syncthreads();
		u32 base = 0, tot = 0;
#pragma unroll
		for (u32 w = 0; w < 16; ++w) {
			const u32 a = s_w[w];
			base += w < wid ? a : 0u;
			tot += a;
		}
		total = tot;
		return base + x - v;
	};
	// the small keys: PER consecutive table entries per thread
	constexpr u32 PER = LOG_NSMALL / 1024;
	u32 sum4 = 0;
#pragma unroll 4
	for (u32 j = 0; j < PER; ++j)
		sum4 += tabs->cnt[PER * tid + j];
	u32 nsmall;
	u32 o = scan1024(sum4, nsmall);
#pragma unroll 4
	for (u32 j = 0; j < PER; ++j) {
		tabs->offs_small[PER * tid + j] = o;
		o += tabs->cnt[PER * tid + j];
	}
	if (tid == 0)
		tabs->offs_small[LOG_NSMALL] = nsmall;
	// the level-1 buckets
	const u32 h = tid < 256 ? tabs->cnt[LOG_NSMALL + tid] : 0u;
	if (tid < 256)
		s_h1[tid] = h;
	u32 nbig, tot1, tot2, ntiles, hmax = h;
	const u32 e_out = scan1024(h, nbig);
	const u32 e_reg = scan1024((h + 7u) & ~7u, tot1);
	const u32 cap2 = h ? log_cap2(h >> 8) : 0u;
	const u32 e_b2 = scan1024(256u * cap2, tot2);
	const u32 nt = (h + tile2 - 1u) / tile2;
	const u32 e_tb = scan1024(nt, ntiles);
#pragma unroll
	for (int off = 32; off > 0; off >>= 1) {
		const u32 y = (u32)__shfl_xor((int)hmax, off);
		hmax = y > hmax ? y : hmax;
	}
	__syncthreads();
	if (lane == 0)
		s_w[wid] = hmax;
	__syncthreads();
	hmax = 0;
#pragma unroll
	for (u32 w = 0; w < 16; ++w)
		hmax = s_w[w] > hmax ? s_w[w] : hmax;
	// (LogCtl::fail names what did not fit: 2 the counts, 4 the level-1 array, 8 the slot array, 16 the tile table, 32 a bucket too
	// large for leaves, 64 a digit the sample's bit length does not have)
	u32 bad = ((u64)nsmall + nbig != n ? 2u : 0u) | (tot1 > l1_cap ? 4u : 0u) | (tot2 > l2_cap ? 8u : 0u) | (ntiles > tiles_cap ? 16u : 0u);
	if (tid < 256)
		bad |= (cap2 > ctl->leaf_cap ? 32u : 0u) | ((h != 0 && tid >= ndig) ? 64u : 0u);
	if (bad)
		atomicOr(&s_fail, bad);
	__syncthreads();
	if (s_fail) {
		if (tid == 0)
			ctl->fail = s_fail;
		return;
	}
	if (tid < 256) {
		tabs->out1[tid] = nsmall + e_out;
		tabs->reg1[tid] = e_reg;
		tabs->cap2[tid] = cap2;
		tabs->base2[tid] = e_b2;
		tabs->tb2[tid] = e_tb;
		for (u32 j = 0; j < nt; ++j) {
			LogTile t;
			t.beg = e_reg + j * tile2;
			t.cnt = h - j * tile2 < tile2 ? h - j * tile2 : tile2;
			t.bucket = tid;
			t.pad = 0;
			tiles[e_tb + j] = t;
		}
	}
	if (tid == 0) {
		tabs->tb2[256] = ntiles;
		// tiles per workgroup of the level-2 pass: a bucket's tiles shared by at most eight workgroups (their carried values fit
		// the 128 places at a slot's end: 8 x 15)
		const u32 maxt = (hmax + tile2 - 1u) / tile2;
		u32 per = (ntiles + grid2 - 1u) / grid2;
		const u32 need = (maxt + 5u) / 6u;
		per = per > need ? per : need;
		ctl->per2 = per ? per : 1u;
		ctl->nsmall = nsmall;
		ctl->ntiles2 = ntiles;
		ctl->maxh1 = hmax;
		ctl->ok = 1u;
	}
	(void)m;
}

// ---- the small keys, written out ---------------------------------------------------------------------------------------------
// out[p] = the element image of value v for offs_small[v] <= p < offs_small[v + 1]: sixteen bytes per lane and step.
constexpr int LOG_FILL_BLOCK = LOG_C <= 12 ? 256 : 1024;   // (the prefix table lies in the LDS: 16 .. 64 KiB per workgroup)
template <typename KT>
__global__ __launch_bounds__(LOG_FILL_BLOCK) void rsx_log_fill_kernel(KT *__restrict__ src, KT *__restrict__ aux,
                                                          const LogCtl *__restrict__ ctl, const LogTabs *__restrict__ tabs,
                                                          KdfArgs<KT> ka)
{
	if (ctl->ok != 1u || ctl->fail)
		return;
	KT *const out = (ctl->ncols & 1u) ? aux : src;   // radix_sort.hpp:92
	__shared__ u32 offs[LOG_NSMALL + 1];
	const u32 tid = threadIdx.x;
	const u32 nsmall = ctl->nsmall;
	constexpr u32 FB = LOG_FILL_BLOCK;
	constexpr u32 CHUNK = FB * 2 * 16;   // positions per workgroup and trip
	if ((u64)blockIdx.x * CHUNK >= nsmall)
		return;
	for (u32 i = tid; i <= LOG_NSMALL; i += FB)
		offs[i] = tabs->offs_small[i];
	__syncthreads();
	const u64 up = (((u64)ctl->key0_hi << 32) | ctl->key0_lo) & ~(((u64)1 << ctl->B) - 1u);
	auto value_at = [&](const u32 p) -> u32 {   // the last v with offs[v] <= p
		u32 lo = 0, hi = LOG_NSMALL;
		while (hi - lo > 1) {
			const u32 mid = (lo + hi) >> 1;
			if (offs[mid] <= p)
				lo = mid;
			else
				hi = mid;
		}
		return lo;
	};
	for (u64 c0 = (u64)blockIdx.x * CHUNK; c0 < nsmall; c0 += (u64)gridDim.x * CHUNK) {
#pragma unroll 1
		for (u32 j = 0; j < 16; ++j) {
			const u64 p = c0 + (u64)(j * FB + tid) * 2;
			if (p >= nsmall)
				break;
			const u32 v0 = value_at((u32)p);
			const u32 v1 = p + 1 < nsmall ? (offs[v0 + 1] > (u32)p + 1 ? v0 : value_at((u32)p + 1)) : v0;
			KT kk[2];
			kk[0] = kdf_invert((KT)(up | v0), ka);
			kk[1] = kdf_invert((KT)(up | v1), ka);
			if (p + 1 < nsmall)
				store_chunk<KT, 2>(out + p, kk);
			else
				out[p] = kk[0];
		}
	}
}

// ---- level 1 ------------------------------------------------------------------------------------------------------------------
// rsx_pass32a_kernel's scheme (rsx_pass32.hpp) with exact buckets: a workgroup owns a contiguous range of 14 Ki-key tiles and carries,
// per digit, the up to seven keys that do not fill a 64-byte atom; atoms go to the bucket's front (one returning global atomic per
// tile and digit on its cursor), what is still carried when the range ends to the bucket's back, counted down from its end.  The
// small keys are passed over.
struct LogP1Cfg {
	static constexpr int BLOCK = 1024, KPT = 14, TILE = BLOCK * KPT, SB = 7;
	static constexpr u32 ATOM = 8, VEC = 2;
	static constexpr int STAGE = TILE + 256 * 2;
};
struct LogP1Smem {
	__attribute__((aligned(16))) u64 stage[LogP1Cfg::STAGE];
	__attribute__((aligned(16))) u64 carry[256][8];
	u32 cell[2][256];
	u32 delta[256];
	u32 info[256];
	unsigned short rbeg[256], bbeg[256], bend[256];
	unsigned char group_digit[LogP1Cfg::STAGE / 2];
	u32 wsum[4];
};

template <typename KT>
__global__ __launch_bounds__(LogP1Cfg::BLOCK, 4) void rsx_log_pass1_kernel(const KT *__restrict__ kin, u64 n, KT *__restrict__ kout,
                                                                          LogCtl *__restrict__ ctl, LogTabs *__restrict__ tabs,
                                                                          KdfArgs<KT> ka)
{
	typedef LogP1Cfg C;
	constexpr int BLOCK = C::BLOCK, KPT = C::KPT, TILE = C::TILE, SB = C::SB;
	constexpr u32 VEC = C::VEC, ATOM = C::ATOM;
	if (ctl->ok != 1u || ctl->fail)
		return;
	const u32 ntiles = (u32)((n + TILE - 1) / TILE);
	const u32 per = (ntiles + gridDim.x - 1) / gridDim.x;
	const u32 t0 = blockIdx.x * per, t1 = t0 + per < ntiles ? t0 + per : ntiles;
	if (t0 >= t1)
		return;
	const u32 m = ctl->m;
	const u64 lowmask = ((u64)1 << ctl->B) - 1u;
	__shared__ LogP1Smem sm;
	const u32 tid0 = threadIdx.x;
	auto sidx = [](u32 pos) { return stage_swz<true>(pos * 8u); };
	auto staged = [&](u32 pos) -> u64 & { return *(u64 *)((char *)sm.stage + sidx(pos)); };
	u32 cc = 0;
	if (tid0 < 256)
		sm.cell[0][tid0] = 0;
	__syncthreads();
	KT keep[KPT];
	auto request = [&](const u32 t, const u32 tid) {
		const u64 beg = (u64)t * TILE;
		const u32 cnt = n - beg < (u64)TILE ? (u32)(n - beg) : (u32)TILE;
		const KT *p = kin + beg;
		if (cnt == (u32)TILE) {
			typedef KT vec_t __attribute__((ext_vector_type(2)));
			const vec_t *vp = (const vec_t *)p + tid;
#pragma unroll
			for (int i = 0; i < KPT / 2; ++i) {
				const vec_t v = vp[i * BLOCK];
				keep[2 * i] = v[0];
				keep[2 * i + 1] = v[1];
			}
		} else {
#pragma unroll
			for (int r = 0; r < KPT; ++r) {
				const u32 o = tid + r * BLOCK;
				keep[r] = o < cnt ? p[o] : (KT)0;
			}
		}
	};
	// (a key's place in a partial tile: element tid + r * BLOCK; in a whole tile any lane holds any key)
	for (u32 t = t0; t < t1; ++t) {
		u32 tid = tid0;
		asm volatile("" : "+v"(tid));
		const u32 lane = tid & 63, wid = tid >> 6;
		const u32 cd = tid >> 2, part = tid & 3u;
		u32 *const cell = sm.cell[(t - t0) & 1u];
		const u64 beg = (u64)t * TILE;
		const u32 cnt = n - beg < (u64)TILE ? (u32)(n - beg) : (u32)TILE;
		const bool full = cnt == (u32)TILE;
		request(t, tid);
		u32 rk[(KPT + 1) / 2];
		auto count = [&](auto full_c) {
			constexpr bool FULL = decltype(full_c)::value;
#pragma unroll
			for (int r = 0; r < KPT; ++r) {
				u32 mine = 0;
				if (FULL || tid + r * BLOCK < cnt) {
					const u64 kk = (u64)kdf_apply(keep[r], ka) & lowmask;
					if (kk >= LOG_NSMALL)
						mine = __hip_atomic_fetch_add(&cell[log_digit(kk, m) & 255u], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
				}
				rk[r >> 1] = (r & 1) ? rk[r >> 1] | (mine << 16) : mine;
			}
		};
		if (full)
			count(std::true_type{});
		else
			count(std::false_type{});
		__syncthreads();

		u32 base = 0;
		{
			u32 rlen = 0, rstart = 0;
			if (tid < 256) {
				const u32 c = cell[tid];
				u32 h, body = 0, tail = 0, atom = 0;
				const bool enough = cc + c >= ATOM;
				if (enough) {
					h = cc ? ATOM - cc : 0u;
					atom = cc ? 1u : 0u;
					body = (c - h) & ~(ATOM - 1u);
					tail = (c - h) & (ATOM - 1u);
				} else {
					h = c;
				}
				const u32 mm = atom * ATOM + body;
				if (mm)
					base = __hip_atomic_fetch_add(&tabs->cur1[tid], mm, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				const u32 o = (VEC - (h & (VEC - 1u))) & (VEC - 1u);
				rlen = (o + c + VEC - 1u) & ~(VEC - 1u);
				sm.info[tid] = cc | (h << 5) | (tail << 10) | (atom << 15) | ((enough ? 1u : 0u) << 16) | (o << 17);
				sm.bend[tid] = (unsigned short)body;
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
				sm.cell[((t - t0) & 1u) ^ 1u][tid] = 0;
				sm.rbeg[tid] = (unsigned short)rb;
				sm.bbeg[tid] = (unsigned short)bb;
				sm.bend[tid] = (unsigned short)be;
				for (u32 g = bb / VEC; g < (be + VEC - 1u) / VEC; ++g)
					sm.group_digit[g] = (unsigned char)tid;
			}
		}
		__syncthreads();
		if (tid < 256) {
			const u32 atom = (sm.info[tid] >> 15) & 1u, bb = sm.bbeg[tid], mm = atom * ATOM + (sm.bend[tid] - bb);
			const u32 dest = tabs->reg1[tid] + base + atom * ATOM;   // of the body's first key
			if (mm && base + mm > tabs->cnt[LOG_NSMALL + tid])
				atomicOr(&ctl->fail, 256u);   // (cannot happen: the buckets are exact)
			sm.delta[tid] = dest - bb;
		}
		u32 m_b = m;
		asm volatile("" : "+s"(m_b));
		auto stage_keys = [&](auto full_c) {
			constexpr bool FULL = decltype(full_c)::value;
#pragma unroll
			for (int r0 = 0; r0 < KPT; r0 += SB) {
				u32 pos[SB];
				bool big[SB];
#pragma unroll
				for (int r = 0; r < SB; ++r) {
					pos[r] = 0;
					big[r] = false;
					if (FULL || tid + (r0 + r) * BLOCK < cnt) {
						const u64 kk = (u64)kdf_apply(keep[r0 + r], ka) & lowmask;
						big[r] = kk >= LOG_NSMALL;
						if (big[r])
							pos[r] = cell[log_digit(kk, m_b) & 255u] + ((rk[(r0 + r) >> 1] >> (16 * ((r0 + r) & 1))) & 0xFFFFu);
					}
				}
#pragma unroll
				for (int r = 0; r < SB; ++r) {
					if (big[r])
						staged(pos[r]) = (u64)keep[r0 + r];
				}
			}
		};
		if (full)
			stage_keys(std::true_type{});
		else
			stage_keys(std::false_type{});
		__syncthreads();
		// ---- out: the completed atoms ...
		{
			const u32 inf = sm.info[cd];
			const u32 ccd = inf & 31u, atomd = (inf >> 15) & 1u;
			if (atomd) {
				const u32 rb = sm.rbeg[cd];
				typedef u64 kvec_t __attribute__((ext_vector_type(2)));
				typedef kvec_t avec_t __attribute__((aligned(16)));
				kvec_t w;
#pragma unroll
				for (u32 e = 0; e < VEC; ++e) {
					const u32 k = part * VEC + e;
					w[e] = k < ccd ? sm.carry[cd][k] : staged(rb + (k - ccd));
				}
				*(avec_t *)((u64 *)kout + (u32)(sm.delta[cd] + sm.bbeg[cd] - ATOM + part * VEC)) = w;
			}
		}
		// ... and the bodies
		{
			const u32 total = (u32)__builtin_amdgcn_readfirstlane((int)sm.wsum[0]) + sm.wsum[1] + sm.wsum[2] + sm.wsum[3];
#pragma unroll 1
			for (u32 i0 = VEC * tid; i0 < total; i0 += VEC * BLOCK) {
				const u32 d = sm.group_digit[i0 / VEC];
				if (i0 >= sm.bbeg[d] && i0 < sm.bend[d]) {
					typedef u64 kvec_t __attribute__((ext_vector_type(2)));
					typedef kvec_t avec_t __attribute__((aligned(16)));
					*(avec_t *)((u64 *)kout + (u32)(sm.delta[d] + i0)) = *(const kvec_t *)((const char *)sm.stage + sidx(i0));
				}
			}
		}
		// ---- what stays
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
	// ---- what is still carried goes to the back of its bucket
	{
		const u32 tid = tid0, cd = tid >> 2, part = tid & 3u;
		__syncthreads();
		if (tid < 256) {
			u32 inf = 0, dest = 0;
			if (cc) {
				const u32 pos = __hip_atomic_fetch_add(&tabs->cur1[256u + tid], cc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				const u32 h1 = tabs->cnt[LOG_NSMALL + tid];
				if (pos + cc > h1)
					atomicOr(&ctl->fail, 512u);
				else {
					inf = cc;
					dest = tabs->reg1[tid] + h1 - pos - cc;
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
				((u64 *)kout)[dest + k] = sm.carry[cd][k];
		}
	}
}

// ---- level 2 ------------------------------------------------------------------------------------------------------------------
// The pass of rsx_pass2w.hpp (8-byte keys in, four-byte values out, whole atoms, two cursors per slot) with this route's tables:
// the tile table of the plan kernel, a shift and a slot capacity per bucket, LogCtl::fail as the verdict.
typedef Pass2wCfg<u32> LogP2Cfg;
static_assert(LogP2Cfg::BACK == LOG_BACK2, "the leaves read a slot's back where the pass writes it");

template <typename KT> struct LogPass2Policy {
	const KT *kin;
	const LogTile *tiles;
	LogCtl *ctl;
	const LogTabs *tabs;
	u32 *cur2;
	u32 dump_at, m;
	__device__ __forceinline__ bool go() const { return ctl->ok == 1u && ctl->fail == 0u; }
	__device__ __forceinline__ u32 ntiles() const { return ctl->ntiles2; }
	__device__ __forceinline__ u32 per(u32) const { return ctl->per2; }
	__device__ __forceinline__ Pass2wTile<KT> tile(u32 t) const
	{
		const LogTile lt = tiles[t];
		return Pass2wTile<KT>{kin + lt.beg, lt.cnt, lt.bucket};
	}
	__device__ __forceinline__ u32 shift(u32 bucket) const { return log_shift2(bucket, m); }
	__device__ __forceinline__ u32 cap(u32 bucket) const { return tabs->cap2[bucket]; }
	__device__ __forceinline__ u32 slot(u32 bucket, u32 d) const { return tabs->base2[bucket] + d * tabs->cap2[bucket]; }
	__device__ __forceinline__ u32 *front(u32 bucket, u32 d) const { return cur2 + bucket * 256u + d; }
	__device__ __forceinline__ u32 *back(u32 bucket, u32 d) const { return cur2 + 65536u + bucket * 256u + d; }
	__device__ __forceinline__ void lost(u32 what) const { atomicOr(&ctl->fail, what == 1u ? 1024u : 2048u); }
	__device__ __forceinline__ u32 dump() const { return dump_at; }
};

// (the level-1 array holds the caller's element images: the pass derives them again; the bits above B do not reach the digit or
// the value -- the digit lies below bit 44, the value is the low word)
template <typename KT>
__global__ __launch_bounds__(LogP2Cfg::BLOCK, 8) void rsx_log_pass2_kernel(const KT *__restrict__ kin, u32 *__restrict__ kout,
                                                                          const LogTile *__restrict__ tiles,
                                                                          LogCtl *__restrict__ ctl, const LogTabs *__restrict__ tabs,
                                                                          u32 *__restrict__ cur2, u32 dump, KdfArgs<KT> ka)
{
	__shared__ Pass2wSmem<u32> sm;
	const LogPass2Policy<KT> pol{kin, tiles, ctl, tabs, cur2, dump, ctl->m};
	pass2w_body<KT, u32, false>(pol, kout, ka, sm);
}

// ---- the leaves ---------------------------------------------------------------------------------------------------------------
// One workgroup per level-2 slot (rsx_leafk_kernel's SLOT32 form, rsx_leaf16.hpp): the slot's values from both of its ends, placed
// by the top twelve of their undecided bits (all of them, if there are at most twelve: the placement then is the sort), finished
// by Batcher's network on sixteen values per lane (or, bins too full for that, over the whole leaf in the LDS); where the leaf
// begins in the sorted array = its bucket's start + the sizes of the slots before it.
template <int BLOCK_, int CAP_, int NBITS_> struct LogLeafCfgT {
	static constexpr int BLOCK = BLOCK_, CAP = CAP_, NW = BLOCK_ / 64, NBITS = NBITS_, NBIN = 1 << NBITS_, NCELLW = NBIN / 2;
	static constexpr int PLANES = NCELLW / 4 / BLOCK;   // 16-byte vectors of cells per thread
	static constexpr int NK = CAP / BLOCK, NCH = (CAP / 16 + BLOCK - 1) / BLOCK;
	static constexpr int S = CAP / 16 + 3;
	static constexpr int WPE = BLOCK == 256 ? 5 : 4;    // (29 KiB / 58 KiB of LDS per workgroup: five / two per CU)
	static_assert(S % 2 == 1, "rows that start in different banks");
	static_assert(PLANES == 2 && NCELLW == 4 * BLOCK * PLANES && CAP % BLOCK == 0, "two vectors of cells per thread");
};
typedef LogLeafCfgT<256, LOG_LEAF_CAP, 12> LogLeafCfg;
typedef LogLeafCfgT<512, LOG_LEAF_CAP_BIG, 13> LogLeafCfgBig;

template <typename KT, typename C = LogLeafCfg>
__global__ __launch_bounds__(C::BLOCK, C::WPE) void rsx_log_leaf_kernel(KT *__restrict__ src, KT *__restrict__ aux,
                                                                           const u32 *__restrict__ slots,
                                                                           const LogCtl *__restrict__ ctl,
                                                                           const LogTabs *__restrict__ tabs,
                                                                           const u32 *__restrict__ cur2, KdfArgs<KT> ka,
                                                                           u32 d1_lo = 0, u32 d1_hi = 256)   // (the probe: a range of level-1 digits)
{
	constexpr int BLOCK = C::BLOCK, NK = C::NK, NCH = C::NCH, NCELLW = C::NCELLW, NW = C::NW, PLANES = C::PLANES, S = C::S;
	constexpr u32 NBITS = C::NBITS;
	// everything the leaf needs to find its values is requested at once: the control block, the slot's two cursors, those of the
	// slots before it in its bucket, the bucket's row of the tables (one round trip in front of the values' instead of four)
	const u32 d1 = blockIdx.x >> 8, d2 = blockIdx.x & 255u;
	const u32 tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
	const u32 ok = ctl->ok, lost = ctl->fail, ndig = ctl->ndig, m = ctl->m, B = ctl->B, ncols = ctl->ncols, shape = ctl->leaf_cap;
	const u64 key0 = ((u64)ctl->key0_hi << 32) | ctl->key0_lo;
	const u32 front = cur2[blockIdx.x], back = cur2[65536u + blockIdx.x], cnt = front + back;
	u32 c_before = tid < d2 ? cur2[d1 * 256u + tid] + cur2[65536u + d1 * 256u + tid] : 0u;
	const u32 cap = tabs->cap2[d1], base2 = tabs->base2[d1], out1 = tabs->out1[d1];
	if (ok != 1u || lost || shape != (u32)C::CAP || d1 >= ndig || d1 < d1_lo || d1 >= d1_hi || cnt == 0 || cnt > (u32)C::CAP)
		return;   // (cnt > CAP cannot happen: a slot that would hold more has set LogCtl::fail; shape: the other launch's leaves)
	const u32 *q = slots + base2 + d2 * cap;
	__shared__ __attribute__((aligned(16))) u32 cell[NCELLW + 64];
	__shared__ __attribute__((aligned(16))) u32 stage[16 * S + 64];
	__shared__ u32 ws[NW], wmax[NW], wpre[NW];
	const u32 blen = log_blen(d1, m), s2 = log_shift2(d1, m);
	const u32 nb = s2 < NBITS ? s2 : NBITS, bsh = s2 - nb, bmask = (1u << nb) - 1u;
	auto at = [](u32 p) { return (p & 15u) * (u32)S + (p >> 4); };
	u32 kv[NK];
#pragma unroll
	for (int j = 0; j < NK; ++j) {
		const u32 e = tid + BLOCK * j;
		kv[j] = e < front ? q[e] : e < cnt ? q[cap - LOG_BACK2 + (e - front)] : 0u;
	}
	// (the sizes of the slots before this one in its bucket, summed: through the first barrier the leaf has anyway)
#pragma unroll
	for (int o = 32; o > 0; o >>= 1)
		c_before += (u32)__shfl_xor((int)c_before, o);
	if (lane == 0)
		wpre[wid] = c_before;
	{
		const u32x4 zero = {0, 0, 0, 0};
#pragma unroll
		for (int j = 0; j < PLANES; ++j)
			((u32x4 *)cell)[tid + BLOCK * j] = zero;
	}
	__syncthreads();
	// (fewer than twelve undecided bits: every value has 2^rep bins, one per lane class -- equal values are equal keys, any order
	// among them will do, and the lanes of a wave that hold the same value no longer queue at one LDS word: the leaves of bit
	// lengths 13 .. 16 of 2^28 Zipf-like keys, two to sixteen values per leaf, took 0.165 ms against 0.093 for the next four)
	const u32 rep = NBITS - nb, repmask = (rep < 6u ? (1u << rep) : 64u) - 1u;   // (a lane class per bin copy: at most the wave's 64)
	auto cell_of = [&](u32 v, bool valid, u32 &sh) -> u32 * {
		const u32 bin = (((v >> bsh) & bmask) << rep) | (lane & repmask);
		sh = (bin & 1u) << 4;
		return &cell[valid ? bin >> 1 : NCELLW + lane];
	};
#pragma unroll
	for (int j = 0; j < NK; ++j) {
		if (BLOCK * j < (int)cnt) {
			u32 sh;
			u32 *a = cell_of(kv[j], tid + BLOCK * j < cnt, sh);
			__hip_atomic_fetch_add(a, 1u << sh, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
		}
	}
	__syncthreads();
	u32x4 c[PLANES];
	u32 pk = 0, mxp = 0;
#pragma unroll
	for (int j = 0; j < PLANES; ++j) {
		c[j] = ((const u32x4 *)cell)[tid + BLOCK * j];
		u32 run = 0;
#pragma unroll
		for (int i = 0; i < 4; ++i) {
			const u32 x = c[j][i];
			mxp = pk_max_u16(mxp, x);
			const u32 lo16 = x & 0xFFFFu, hs = run + lo16;
			c[j][i] = run | (hs << 16);
			run = hs + (x >> 16);
		}
		pk |= run << (16 * j);
	}
	u32 mx = (mxp & 0xFFFFu) > (mxp >> 16) ? (mxp & 0xFFFFu) : (mxp >> 16);
	const u32 incl = wave_incl_scan_dpp(pk);
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) {
		const u32 y = (u32)__shfl_xor((int)mx, o);
		mx = mx > y ? mx : y;
	}
	if (lane == 63) {
		ws[wid] = incl;
		wmax[wid] = mx;
	}
	__syncthreads();
	mx = wmax[0];
#pragma unroll
	for (int w = 1; w < NW; ++w)
		mx = mx > wmax[w] ? mx : wmax[w];
	{
		u32 base = 0, tot = 0;
#pragma unroll
		for (u32 w = 0; w < (u32)NW; ++w) {
			const u32 a = ws[w];
			base += w < wid ? a : 0u;
			tot += a;
		}
		const u32 e = incl - pk + base;
		const u32 o[2] = {e & 0xFFFFu, (tot & 0xFFFFu) + (e >> 16)};
#pragma unroll
		for (int j = 0; j < PLANES; ++j) {
			const u32 bb = o[j] | (o[j] << 16);
			u32x4 x;
#pragma unroll
			for (int i = 0; i < 4; ++i)
				x[i] = c[j][i] + bb;
			((u32x4 *)cell)[tid + BLOCK * j] = x;
		}
	}
	__syncthreads();
#pragma unroll
	for (int j = 0; j < NK; ++j) {
		if (BLOCK * j < (int)cnt) {
			const bool valid = tid + BLOCK * j < cnt;
			u32 sh;
			u32 *a = cell_of(kv[j], valid, sh);
			const u32 old = __hip_atomic_fetch_add(a, 1u << sh, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
			const u32 pos = (old >> sh) & 0xFFFFu;
			stage[valid ? at(pos) : 16 * S + lane] = kv[j];
		}
	}
	if (tid < 32) {
		const u32 p = cnt + tid;
		stage[at(p)] = ~0u;   // what the last chunks read behind the leaf's end sorts last
	}
	__syncthreads();
	if (bsh != 0) {   // (more than twelve undecided bits: the bins hold different values)
		if (mx > 25u)
			batcher_sort_lds<BLOCK>(stage, cnt, at);
		const u32 npass = mx > 25u ? 0u : mx > 17u ? 4u : mx > 9u ? 3u : 2u;
		for (u32 pass = 0; pass < npass; ++pass) {
			const u32 off = 8 * (pass & 1);
#pragma unroll
			for (int r = 0; r < NCH; ++r) {
				const u32 ch = tid + BLOCK * r;
				if (16 * ch + off < cnt) {
					u32 d[16];
#pragma unroll
					for (int i = 0; i < 16; ++i)
						d[i] = (pass & 1) ? (i < 8 ? stage[(i + 8) * S + ch] : stage[(i - 8) * S + ch + 1]) : stage[i * S + ch];
					if (pass == 0)
						sort16_values(d);
					else
						merge16_values(d);
#pragma unroll
					for (int i = 0; i < 16; ++i) {
						if (pass & 1) {
							if (i < 8)
								stage[(i + 8) * S + ch] = d[i];
							else
								stage[(i - 8) * S + ch + 1] = d[i];
						} else {
							stage[i * S + ch] = d[i];
						}
					}
				}
			}
			__syncthreads();
		}
	}
	{
		// what every key of the leaf has above its low word: the first key's constant top bits, the leading one, the mantissa
		// bits and the level-2 digit where they lie above bit 32
		const u64 lowmask = ((u64)1 << B) - 1u;
		u64 up = key0 & ~lowmask;
		up |= (((u64)1 << (blen - 1u)) | ((u64)(d1 & ((1u << m) - 1u)) << (blen - 1u - m)) | ((u64)d2 << s2)) & ~(u64)0xFFFFFFFFu;
		u32 pre = 0;
#pragma unroll
		for (int w = 0; w < NW; ++w)
			pre += wpre[w];
		KT *o = ((ncols & 1u) ? aux : src) + out1 + pre;   // radix_sort.hpp:92
		for (u32 i0 = 2 * tid; i0 < cnt; i0 += 2 * BLOCK) {
			KT kk[2];
			kk[0] = kdf_invert((KT)(up | stage[at(i0)]), ka);
			kk[1] = kdf_invert((KT)(up | stage[at(i0 + 1)]), ka);
			if (i0 + 2 <= cnt)
				store_chunk<KT, 2>(o + i0, kk);
			else
				o[i0] = kk[0];
		}
	}
}

}  // namespace rsx
