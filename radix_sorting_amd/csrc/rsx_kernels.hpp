// rsx_kernels.hpp -- CDNA4 (gfx950) kernels of the LSD radix sort.
//
// Three kernels replace the three loops of the reference's rs_sort_main
// (radix_sort.hpp:31-93):
//
//   rsx_hist_kernel     loop 1 (:48-58): every 8-bit column's histogram in ONE
//                       read of the keys + the pre-sorted test.  Histograms are
//                       privatised in LDS per workgroup (R lane-striped copies
//                       per bin so equal digits do not serialise on one LDS
//                       address) and merged into HBM with global atomics.
//   rsx_plan_kernel     column-skip probe (:64-70) + exclusive scan (:72-80):
//                       one 64-lane wavefront per column, 4 bins per lane,
//                       scanned through LDS.
//   rsx_scatter_kernel  one scatter pass (:82-90) as a single-read/single-write
//                       "onesweep": a workgroup takes a tile (ticket order),
//                       ranks its keys inside each wavefront with 8 ballots +
//                       mbcnt/popcount, chains the per-digit tile offsets with a
//                       decoupled look-back over agent-scope status words, stages
//                       the tile in LDS in output order and writes coalesced runs.
//
// Stability: a wave owns a contiguous slice of the tile, lane l of round r holds
// element slice + 64 r + l, rounds are ranked in order and lanes by mbcnt, waves
// and tiles are prefix-summed in memory order -- so equal digits keep their
// input order exactly as the reference's in-order traversal with post-increment
// does.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rsx {

typedef unsigned long long u64;
typedef unsigned int u32;

struct NoVal {};  // "keys only" payload tag

template <typename T> struct val_bytes { static constexpr int value = sizeof(T); };
template <> struct val_bytes<NoVal> { static constexpr int value = 0; };

// ---- key derivation ---------------------------------------------------------
// radix_sort_basic_kdf.hpp:19-46 folded into three per-launch constants so one
// kernel serves unsigned / signed / float keys and both orders:
//   unsigned  fmask = 0   sflip = 0        kdf = k
//   signed    fmask = 0   sflip = highbit  kdf = k ^ highbit            (:26-30)
//   float     fmask = ~0  sflip = highbit  kdf = k ^ (-(k>>31) | 1<<31) (:32-46)
//   desc = ~0 complements the result (README.md:564-574).
template <typename KT> struct KdfArgs { KT fmask, sflip, desc; };

template <typename KT>
__device__ __forceinline__ KT kdf_apply(KT raw, const KdfArgs<KT> a)
{
	typedef typename std::make_signed<KT>::type ST;
	const KT sign = (KT)((ST)raw >> (sizeof(KT) * 8 - 1));  // all ones iff top bit set
	return (KT)(raw ^ ((sign & a.fmask) | a.sflip) ^ a.desc);
}

__device__ __forceinline__ u32 lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

// popcount of the bits of `m` below this lane
__device__ __forceinline__ u32 mbcnt64(u64 m)
{
	return __builtin_amdgcn_mbcnt_hi((u32)(m >> 32), __builtin_amdgcn_mbcnt_lo((u32)m, 0u));
}

// =============================================================================
// Kernel 1: histogram of all columns + pre-sorted test
// =============================================================================

template <typename KT> struct HistCfg {
	static constexpr int WC = sizeof(KT);               // columns
	static constexpr int VEC = 16 / sizeof(KT);         // elements per 16-byte lane load
	static constexpr int R = sizeof(KT) == 8 ? 4 : 8;   // lane-striped copies per bin
	static constexpr int BLOCK = 256;
};

template <typename KT, int R>
__device__ __forceinline__ void hist_add_one(u32 *lh, KT k, u32 lane)
{
#pragma unroll
	for (int j = 0; j < (int)sizeof(KT); ++j) {
		const u32 d = (u32)(k >> (8 * j)) & 0xFFu;
		atomicAdd(&lh[(j * 256 + d) * R + (lane & (R - 1))], 1u);
	}
}

template <typename KT>
__global__ __launch_bounds__(256) void rsx_hist_kernel(const KT *__restrict__ src, u64 n, u64 *__restrict__ ghist,
                                                       u32 *__restrict__ unsorted, KdfArgs<KT> ka)
{
	typedef HistCfg<KT> C;
	constexpr int WC = C::WC, VEC = C::VEC, R = C::R;
	__shared__ u32 lh[WC * 256 * R];
	const u32 tid = threadIdx.x;
	const u32 lane = tid & 63;
	for (u32 i = tid; i < WC * 256 * R; i += C::BLOCK)
		lh[i] = 0;
	__syncthreads();

	// elements before the first 16-byte boundary, and after the last full vector
	const uintptr_t addr = (uintptr_t)src;
	u64 head = ((16 - (addr & 15)) & 15) / sizeof(KT);
	if (head > n)
		head = n;
	const u64 nvec = (n - head) / VEC;
	const u64 tail_begin = head + nvec * VEC;
	bool descent = false;

	if (blockIdx.x == 0) {
		// scalar fringe: < 2*VEC elements in total
		for (u64 i = tid; i < head + (n - tail_begin); i += C::BLOCK) {
			const u64 e = i < head ? i : tail_begin + (i - head);
			const KT k = kdf_apply(src[e], ka);
			if (e + 1 < n && k > kdf_apply(src[e + 1], ka))
				descent = true;
			hist_add_one<KT, R>(lh, k, lane);
		}
	}

	typedef KT vec_t __attribute__((ext_vector_type(VEC)));
	const vec_t *vsrc = (const vec_t *)(src + head);
	constexpr int U = 4;   // independent 16-byte loads in flight per lane
	const u64 stride = (u64)gridDim.x * (C::BLOCK * U);
	for (u64 v0 = (u64)blockIdx.x * (C::BLOCK * U) + tid; v0 < nvec; v0 += stride) {
		vec_t raw[U];
#pragma unroll
		for (int u = 0; u < U; ++u) {
			const u64 v = v0 + (u64)u * C::BLOCK;
			if (v < nvec)
				raw[u] = vsrc[v];
		}
#pragma unroll
		for (int u = 0; u < U; ++u) {
			const u64 v = v0 + (u64)u * C::BLOCK;
			if (v >= nvec)
				break;
			KT k[VEC];
#pragma unroll
			for (int e = 0; e < VEC; ++e)
				k[e] = kdf_apply((KT)raw[u][e], ka);

			// pre-sorted test (radix_sort.hpp:51-54): inside the vector, then against the
			// next element, which the next lane holds except at the wave's right edge.
#pragma unroll
			for (int e = 0; e + 1 < VEC; ++e)
				descent |= k[e] > k[e + 1];
			KT nxt;
			if (sizeof(KT) == 8) {
				const u32 lo = __shfl_down((u32)k[0], 1), hi = __shfl_down((u32)((u64)k[0] >> 32), 1);
				nxt = (KT)(((u64)hi << 32) | lo);
			} else {
				nxt = (KT)__shfl_down((u32)k[0], 1);
			}
			const u64 next_elem = head + (v + 1) * VEC;
			const bool edge = lane == 63 || v + 1 >= nvec;  // the next lane is idle or holds another row
			if (edge)
				nxt = next_elem < n ? kdf_apply(src[next_elem], ka) : k[VEC - 1];
			descent |= k[VEC - 1] > nxt;

			// histogram.  A column whose digit is identical across the whole wave (the
			// column-skip case, radix_sort.hpp:64-70) is counted by one lane.
			const KT first = (KT)(sizeof(KT) == 8
			                          ? (((u64)__builtin_amdgcn_readfirstlane((u32)((u64)k[0] >> 32)) << 32) |
			                             __builtin_amdgcn_readfirstlane((u32)k[0]))
			                          : __builtin_amdgcn_readfirstlane((u32)k[0]));
			KT diff = 0;
#pragma unroll
			for (int e = 0; e < VEC; ++e)
				diff |= (KT)(k[e] ^ first);
			const u64 active = __ballot(1);
#pragma unroll
			for (int j = 0; j < WC; ++j) {
				const bool differs = ((u32)(diff >> (8 * j)) & 0xFFu) != 0;
				if (__any(differs)) {
#pragma unroll
					for (int e = 0; e < VEC; ++e) {
						const u32 d = (u32)(k[e] >> (8 * j)) & 0xFFu;
						atomicAdd(&lh[(j * 256 + d) * R + (lane & (R - 1))], 1u);
					}
				} else if (mbcnt64(active) == 0) {
					const u32 d = (u32)(first >> (8 * j)) & 0xFFu;
					atomicAdd(&lh[(j * 256 + d) * R], (u32)(VEC * __popcll(active)));
				}
			}
		}
	}

	if (__any(descent) && mbcnt64(__ballot(1)) == 0)
		atomicOr(unsorted, 1u);

	__syncthreads();
	for (u32 i = tid; i < WC * 256; i += C::BLOCK) {
		u32 s = 0;
#pragma unroll
		for (int r = 0; r < R; ++r)
			s += lh[i * R + r];
		if (s)
			atomicAdd(&ghist[i], (u64)s);
	}
}

// =============================================================================
// Kernel 2: column-skip probe + exclusive scan (one wavefront per column)
// =============================================================================

struct Plan {            // 64 bytes, copied to the host after this kernel
	u32 ncols;           // kept columns (radix_sort.hpp:64-70)
	u32 sorted;          // 1: pre-sorted early exit (radix_sort.hpp:60-62)
	u32 cols[8];
	u32 pad[6];
};

// Launched with one workgroup of WC wavefronts.  ghist holds counts on entry and
// exclusive offsets on exit, like the reference's histogram array.
template <typename KT>
__global__ __launch_bounds__(64 * sizeof(KT)) void rsx_plan_kernel(const KT *__restrict__ src, u64 n, u64 *__restrict__ ghist,
                                                                  const u32 *__restrict__ unsorted, Plan *__restrict__ plan,
                                                                  KdfArgs<KT> ka)
{
	constexpr int WC = sizeof(KT);
	__shared__ u64 lsum[WC][64];
	__shared__ u32 kept[WC];
	const u32 lane = threadIdx.x & 63, col = threadIdx.x >> 6;

	const KT key0 = kdf_apply(src[0], ka);                        // radix_sort.hpp:65
	const u32 d0 = (u32)(key0 >> (8 * col)) & 0xFFu;
	u64 *h = ghist + 256 * col;
	if (lane == 0)
		kept[col] = h[d0] != n;                                   // radix_sort.hpp:67

	// single-wavefront scan of 256 bins: 4 bins per lane, lane totals through LDS
	u64 c[4];
#pragma unroll
	for (int i = 0; i < 4; ++i)
		c[i] = h[4 * lane + i];
	const u64 mine = c[0] + c[1] + c[2] + c[3];
	u64 incl = mine;
	lsum[col][lane] = incl;
	__syncthreads();
#pragma unroll
	for (int off = 1; off < 64; off <<= 1) {                       // Hillis-Steele inside one wave
		const u64 add = lane >= (u32)off ? lsum[col][lane - off] : 0;
		__syncthreads();
		incl += add;
		lsum[col][lane] = incl;
		__syncthreads();
	}
	u64 a = incl - mine;                                           // radix_sort.hpp:74-79
#pragma unroll
	for (int i = 0; i < 4; ++i) {
		h[4 * lane + i] = a;
		a += c[i];
	}

	__syncthreads();
	if (threadIdx.x == 0) {
		u32 nc = 0;
		for (int i = 0; i < WC; ++i)
			if (kept[i])
				plan->cols[nc++] = i;
		plan->ncols = nc;
		plan->sorted = *unsorted == 0;
	}
}

// =============================================================================
// Kernel 3: one stable scatter pass (onesweep)
// =============================================================================

template <typename ST> struct StatusBits;
template <> struct StatusBits<u32> {
	static constexpr u32 SHIFT = 30;
	static constexpr u32 VALMASK = (1u << 30) - 1;
};
template <> struct StatusBits<u64> {
	static constexpr u32 SHIFT = 62;
	static constexpr u64 VALMASK = (1ull << 62) - 1;
};
enum : u32 { ST_EMPTY = 0, ST_AGGREGATE = 1, ST_PREFIX = 2 };

enum : u32 {
	SCATTER_GEN_INDEX = 1,   // payload of element i is i (first rank pass, radix_sort_rank.hpp:52)
	SCATTER_SKIP_KEYS = 2,   // do not write keys (last rank pass: only the indices are wanted)
	SCATTER_USE_LUT = 4      // bucket = lut[digit] (MSD split for the multi-GPU sort)
};

// Lanes holding the same 8-bit digit as this lane (valid lanes only).
__device__ __forceinline__ u64 match_digit(u32 d, bool valid)
{
	u64 m = __ballot(valid);
#pragma unroll
	for (int b = 0; b < 8; ++b) {
		const bool bit = (d >> b) & 1u;
		const u64 bal = __ballot(bit);
		m &= bit ? bal : ~bal;
	}
	return m;
}

template <typename KT, typename VT> struct ScatterCfg {
	static constexpr int NWAVES = 8;
	static constexpr int BLOCK = NWAVES * 64;
	static constexpr int ELEM = sizeof(KT) > (size_t)val_bytes<VT>::value ? sizeof(KT) : val_bytes<VT>::value;
	static constexpr int KPT = ELEM == 8 ? 8 : 16;       // keys per lane: 32 KiB of staging either way
	static constexpr int TILE = BLOCK * KPT;
};

template <typename KT, typename VT> struct ScatterSmem {
	typedef ScatterCfg<KT, VT> C;
	__attribute__((aligned(16))) unsigned char stage_raw[C::TILE * C::ELEM];
	u32 whist[C::NWAVES][256];   // per-wave digit counters, later per-wave bases
	u64 delta[256];              // global offset of a digit's run minus its tile-local offset
	u32 wsum[4];
	u32 tile;
};

template <typename KT, typename VT, typename ST, bool FULL>
__device__ __forceinline__ void scatter_tile(ScatterSmem<KT, VT> &sm, const KT *__restrict__ kin, KT *__restrict__ kout,
                                             const VT *__restrict__ vin, VT *__restrict__ vout, const u32 tile,
                                             const u64 tile_base, const u32 tile_count, const u32 shift,
                                             const u64 *__restrict__ gbase, ST *status, const KdfArgs<KT> ka,
                                             const u32 flags, const uint8_t *__restrict__ lut)
{
	typedef ScatterCfg<KT, VT> C;
	constexpr int NWAVES = C::NWAVES, BLOCK = C::BLOCK, KPT = C::KPT;
	constexpr bool HAS_VAL = val_bytes<VT>::value != 0;
	typedef StatusBits<ST> SB;
	KT *stage_k = (KT *)sm.stage_raw;
	const u32 tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;

	// ---- load: wave w owns [tile_base + w*64*KPT, +64*KPT), lane l of round r the element 64 r + l of it
	const u32 wofs = wid * (64 * KPT) + lane;
	KT key[KPT];
	VT val[KPT];
#pragma unroll
	for (int r = 0; r < KPT; ++r) {
		const u32 o = wofs + r * 64;
		key[r] = (FULL || o < tile_count) ? kin[tile_base + o] : (KT)0;
	}
	if constexpr (HAS_VAL) {
		if (flags & SCATTER_GEN_INDEX) {
#pragma unroll
			for (int r = 0; r < KPT; ++r)
				val[r] = (VT)(tile_base + wofs + r * 64);
		} else {
#pragma unroll
			for (int r = 0; r < KPT; ++r) {
				const u32 o = wofs + r * 64;
				val[r] = (FULL || o < tile_count) ? vin[tile_base + o] : (VT)0;
			}
		}
	}

	// ---- rank inside the wave, rounds in memory order
	u32 rk[KPT];   // bits 0..15 rank, bits 16..23 digit
	u32 *wh = sm.whist[wid];
#pragma unroll
	for (int r = 0; r < KPT; ++r) {
		const bool valid = FULL || (wofs + r * 64 < tile_count);
		u32 d = (u32)(kdf_apply(key[r], ka) >> shift) & 0xFFu;
		if (flags & SCATTER_USE_LUT)
			d = lut[d];
		const u64 m = match_digit(d, valid);
		const u32 cnt = (u32)__popcll(m);
		const u32 below = mbcnt64(m);
		const u32 prev = wh[d];
		if (valid && below == cnt - 1)
			wh[d] = prev + cnt;
		rk[r] = (prev + below) | (d << 16);
	}
	__syncthreads();

	// ---- digit thread d: prefix over waves, publish the tile aggregate, prefix over digits
	u32 tile_cnt = 0, incl = 0;
	ST *my_status = status + (u64)tile * 256 + tid;
	if (tid < 256) {
#pragma unroll
		for (int w = 0; w < NWAVES; ++w) {
			const u32 t = sm.whist[w][tid];
			sm.whist[w][tid] = tile_cnt;
			tile_cnt += t;
		}
		const ST word = ((ST)(tile == 0 ? ST_PREFIX : ST_AGGREGATE) << SB::SHIFT) | (ST)tile_cnt;
		__hip_atomic_store(my_status, word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		incl = tile_cnt;
#pragma unroll
		for (int off = 1; off < 64; off <<= 1) {
			const u32 t = __shfl_up(incl, off);
			if (lane >= (u32)off)
				incl += t;
		}
		if (lane == 63)
			sm.wsum[wid] = incl;
	}
	__syncthreads();
	u32 tbase = 0;
	if (tid < 256) {
		for (u32 w = 0; w < wid; ++w)
			tbase += sm.wsum[w];
		tbase += incl - tile_cnt;   // tile-local start of digit tid's run
#pragma unroll
		for (int w = 0; w < NWAVES; ++w)
			sm.whist[w][tid] += tbase;
	}
	__syncthreads();

	// ---- stage keys in output order
#pragma unroll
	for (int r = 0; r < KPT; ++r) {
		const bool valid = FULL || (wofs + r * 64 < tile_count);
		const u32 d = rk[r] >> 16;
		const u32 pos = wh[d] + (rk[r] & 0xFFFFu);
		rk[r] = pos;
		if (valid)
			stage_k[pos] = key[r];
	}

	// ---- decoupled look-back over the predecessors' status words (digit thread d)
	if (tid < 256) {
		u64 excl = 0;
		if (tile != 0) {
			const ST *p = my_status - 256;
			for (;;) {
				const ST w = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				const u32 f = (u32)(w >> SB::SHIFT);
				if (f == ST_EMPTY) {
					__builtin_amdgcn_s_sleep(1);
					continue;
				}
				excl += (u64)(w & SB::VALMASK);
				if (f == ST_PREFIX)
					break;
				p -= 256;
			}
			const ST word = ((ST)ST_PREFIX << SB::SHIFT) | (ST)(excl + tile_cnt);
			__hip_atomic_store(my_status, word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		}
		sm.delta[tid] = gbase[tid] + excl - tbase;
	}
	__syncthreads();

	// ---- write out: consecutive staged elements of one digit go to consecutive addresses
	u32 dig[KPT];
#pragma unroll
	for (int k = 0; k < KPT; ++k) {
		const u32 i = tid + k * BLOCK;
		if (FULL || i < tile_count) {
			const KT kv = stage_k[i];
			u32 d = (u32)(kdf_apply(kv, ka) >> shift) & 0xFFu;
			if (flags & SCATTER_USE_LUT)
				d = lut[d];
			dig[k] = d;
			if (!(flags & SCATTER_SKIP_KEYS))
				kout[sm.delta[d] + i] = kv;
		}
	}
	if constexpr (HAS_VAL) {
		VT *stage_v = (VT *)sm.stage_raw;
		__syncthreads();
#pragma unroll
		for (int r = 0; r < KPT; ++r) {
			const bool valid = FULL || (wofs + r * 64 < tile_count);
			if (valid)
				stage_v[rk[r]] = val[r];
		}
		__syncthreads();
#pragma unroll
		for (int k = 0; k < KPT; ++k) {
			const u32 i = tid + k * BLOCK;
			if (FULL || i < tile_count)
				vout[sm.delta[dig[k]] + i] = stage_v[i];
		}
	}
}

template <typename KT, typename VT, typename ST>
__global__ __launch_bounds__(512) void rsx_scatter_kernel(const KT *__restrict__ kin, KT *__restrict__ kout,
                                                          const VT *__restrict__ vin, VT *__restrict__ vout, u64 n,
                                                          u32 shift, const u64 *__restrict__ gbase, ST *status,
                                                          u32 *ticket, KdfArgs<KT> ka, u32 flags,
                                                          const uint8_t *__restrict__ lut)
{
	typedef ScatterCfg<KT, VT> C;
	__shared__ ScatterSmem<KT, VT> sm;
	const u32 tid = threadIdx.x;
	if (tid == 0)
		sm.tile = atomicAdd(ticket, 1u);   // tiles are handed out in start order => look-back cannot deadlock
	for (u32 i = tid; i < C::NWAVES * 256; i += C::BLOCK)
		(&sm.whist[0][0])[i] = 0;
	__syncthreads();
	const u32 tile = sm.tile;
	const u64 tile_base = (u64)tile * C::TILE;
	const u32 tile_count = (n - tile_base) < (u64)C::TILE ? (u32)(n - tile_base) : (u32)C::TILE;
	if (tile_count == (u32)C::TILE)
		scatter_tile<KT, VT, ST, true>(sm, kin, kout, vin, vout, tile, tile_base, tile_count, shift, gbase, status, ka,
		                               flags, lut);
	else
		scatter_tile<KT, VT, ST, false>(sm, kin, kout, vin, vout, tile, tile_base, tile_count, shift, gbase, status, ka,
		                                flags, lut);
}

// =============================================================================
// Small helpers
// =============================================================================

// splitmix64 evaluated counter-based: output number k (0-based) of the stream seeded with s
// is mix(s + (k+1) * 0x9E3779B97F4A7C15) (SURVEY.md section 4 generator).
template <typename T>
__global__ void rsx_fill_splitmix_kernel(T *dst, u64 n, u64 seed, u64 mask, u64 first)
{
	const u64 stride = (u64)gridDim.x * blockDim.x;
	for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
		u64 z = seed + (first + i + 1) * 0x9E3779B97F4A7C15ull;
		z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
		z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
		z ^= z >> 31;
		dst[i] = (T)(z & mask);
	}
}

template <typename IT>
__global__ void rsx_iota_kernel(IT *dst, u64 n)
{
	const u64 stride = (u64)gridDim.x * blockDim.x;
	for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
		dst[i] = (IT)i;
}

// dst[i] = (narrower or wider) src[i]
template <typename DT, typename ST_>
__global__ void rsx_convert_kernel(DT *dst, const ST_ *src, u64 n)
{
	const u64 stride = (u64)gridDim.x * blockDim.x;
	for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
		dst[i] = (DT)src[i];
}

// dst record i = src record idx[i]; records are rec_bytes wide, moved in WORD units
template <typename WORD, typename IT>
__global__ void rsx_gather_kernel(WORD *__restrict__ dst, const WORD *__restrict__ src, const IT *__restrict__ idx,
                                  u64 n, u32 words_per_rec)
{
	const u64 total = n * words_per_rec;
	const u64 stride = (u64)gridDim.x * blockDim.x;
	for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
		const u64 rec = i / words_per_rec;
		const u32 w = (u32)(i - rec * words_per_rec);
		dst[i] = src[(u64)idx[rec] * words_per_rec + w];
	}
}

}  // namespace rsx
