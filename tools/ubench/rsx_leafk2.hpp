// rsx_leafk2.hpp -- variants of rsx_leafk_kernel<u64, u64, ...> (csrc/rsx_leaf16.hpp), the leaves of 2^28 uniform u64 keys
// (BASELINE.json cfg 3): round 4 left that kernel at 1455 us = 0.37 of the HBM peak with 20 bytes of scratch per lane.
// Probe only (tools/ubench/leafk_probe.hip); what wins moves into csrc/rsx_leaf16.hpp.
//
// What can vary here:
//   BLOCK   any multiple of 64 (the cells' scan takes ceil(512 / BLOCK) vectors per thread; vectors that do not exist are skipped)
//   LOOP    false: one leaf per workgroup (grid = number of leaves, what the library launches) -- nothing is loop-invariant,
//           so nothing is hoisted in front of a loop that runs once and spilled there
//   PLANES6 the staged values as a 32-bit plane (bits 16 .. 47 of the derived key) and a 16-bit plane (bits 0 .. 15): 6 bytes
//           per key instead of 8 -> 38 KiB of LDS per 5120-key leaf instead of 50
//   SKIP    1 no register passes, 2 no count / scan / placement, 4 no write-out (wrong output: what each phase costs)
#pragma once

#include "rsx_leaf16.hpp"

namespace rsx {

template <int BLOCK_, int CAP_, int WPE_, int NBITS_ = 12, bool LOOP_ = false, int SKIP_ = 0, bool P6_ = false, bool VLOAD_ = false> struct LeafK2Cfg {
	static constexpr int BLOCK = BLOCK_, CAP = CAP_, WPE = WPE_, NW = BLOCK_ / 64, NBITS = NBITS_;
	static constexpr bool LOOP = LOOP_;
	static constexpr int SKIP = SKIP_;
	static constexpr bool VLOAD = VLOAD_;   // the slot's keys with 16-byte loads, two keys per lane (any order will do)
	static constexpr bool P6 = P6_;   // values staged as a 32-bit plane (bits 16 .. 47) and a 16-bit plane (bits 0 .. 15)
	static constexpr int NCH = (CAP / 16 + BLOCK - 1) / BLOCK;
	static constexpr int NBIN = 1 << NBITS, NCELLW = NBIN / 2, NVEC = NCELLW / 4;
	static constexpr int PLANES = (NVEC + BLOCK - 1) / BLOCK;
	static constexpr u32 MAXBIN = 9, MAXBIN2 = 25;
	static constexpr int S = CAP / 16 + 3;
	static_assert(BLOCK % 64 == 0 && CAP % 16 == 0 && CAP <= 8192, "");
	static_assert(PLANES == 1 || PLANES == 2, "one or two vectors of cells per thread");
	static_assert(S % 2 == 1, "rows that start in different banks");
};

template <typename KT, typename CT, typename C>
__global__ __launch_bounds__(C::BLOCK, C::WPE) void rsx_leafk2_kernel(KT *__restrict__ src, KT *__restrict__ aux,
                                                                      const Plan *__restrict__ plan,
                                                                      const LeafSeg *__restrict__ segtab, SegCtl *__restrict__ ctl,
                                                                      KdfArgs<KT> ka, u32 lo, u32 hi, const KT *__restrict__ slots,
                                                                      u32 slack_cap, u32 *__restrict__ redo, u32 maxbin2 = C::MAXBIN2)
{
	static_assert(sizeof(KT) == 8 && (sizeof(CT) == 4 || sizeof(CT) == 8), "8-byte keys carried as 4- or 8-byte values");
	constexpr u32 NB2 = C::NBITS - 8;
	constexpr int BLOCK = C::BLOCK, CAP = C::CAP, NCH = C::NCH, NCELLW = C::NCELLW, NW = C::NW, PLANES = C::PLANES, NVEC = C::NVEC;
	constexpr int NK = (CAP + BLOCK - 1) / BLOCK;
	const u32 hyb = plan->hyb, ncols = plan->ncols;
	const u32 mode = ctl->mode, maxleaf = ctl->maxleaf, nseg = ctl->nleaf, on = ctl->leaf16;
	if (hyb != HYB_TWO_LEVEL || ncols < 4 || mode != SEG_MODE_LEAVES || maxleaf <= lo || maxleaf > hi || !on)
		return;
	if (ctl->narrow != 0u)
		return;
	const u32 c_hi = plan->cols[ncols - 3] & 7u, c_nx = plan->cols[ncols >= 4 ? ncols - 4 : 0] & 7u;
	if ((sizeof(CT) == 4) != (c_hi <= 3u))
		return;
	const u32 sh_hi = 8 * c_hi, sh_nx = 8 * c_nx + 8 - NB2;
	KT *out = (ncols & 1) ? aux : src;
	__shared__ __attribute__((aligned(16))) u32 cell[NCELLW + 64];
	constexpr int S = C::S;
	constexpr bool P6 = C::P6 && sizeof(CT) == 8;
	constexpr int S2 = (S + 3) & ~1;   // (the 16-bit plane's row pitch: an even number of halves, rows start in different banks)
	__shared__ __attribute__((aligned(16))) CT stage[P6 ? 1 : 16 * S + 64];
	__shared__ __attribute__((aligned(16))) u32 st_hi[P6 ? 16 * S + 64 : 1];
	__shared__ __attribute__((aligned(16))) unsigned short st_lo[P6 ? 16 * S2 + 64 : 1];
	auto at = [](u32 p) { return (p & 15u) * (u32)S + (p >> 4); };
	// element (row r, column c) of the staged leaf; the places behind the rows (16 * S + lane) take values that do not exist
	auto put = [&](u32 r, u32 c, CT v) {
		if constexpr (P6) {
			st_hi[r * S + c] = (u32)((u64)v >> 16);
			st_lo[r * S2 + c] = (unsigned short)v;
		} else {
			stage[r * S + c] = v;
		}
	};
	auto get = [&](u32 r, u32 c) -> CT {
		if constexpr (P6)
			return (CT)(((u64)st_hi[r * S + c] << 16) | st_lo[r * S2 + c]);
		else
			return stage[r * S + c];
	};
	auto put_at = [&](u32 p, CT v) { put(p & 15u, p >> 4, v); };
	auto get_at = [&](u32 p) -> CT { return get(p & 15u, p >> 4); };
	__shared__ u32 ws[NW], wmax[NW];
	const u32 tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
	for (u32 s = blockIdx.x; s < nseg; s += gridDim.x) {
		const LeafSeg ls = segtab[s];
		const u32 cnt = ls.cnt, slot = ls.slot;
		if (cnt != 0) {
			const KT *q = slot ? slots + (u64)(slot - 1) * slack_cap : (const KT *)src + ls.beg;
			CT kv[NK];
			const KT first = kdf_apply(q[0], ka);
			// element index of register j (VLOAD: lane t holds elements 2 t, 2 t + 1 of every 2 * BLOCK: a slot starts on a 16-byte
			// boundary and its capacity is even, so the second element of a vector is the slot's own even behind the last key)
			auto elem_of = [&](int j) { return C::VLOAD ? 2u * tid + 2u * BLOCK * (u32)(j >> 1) + (u32)(j & 1) : tid + BLOCK * (u32)j; };
			if constexpr (C::VLOAD) {
				static_assert(NK % 2 == 0, "whole vectors");
				typedef KT kvec_t __attribute__((ext_vector_type(2)));
#pragma unroll
				for (int j = 0; j < NK; j += 2) {
					const u32 e = elem_of(j);
					kvec_t x = {0, 0};
					if (e < cnt)
						x = *(const kvec_t *)(q + e);
					kv[j] = (CT)(P6 ? (kdf_apply(x[0], ka) & (KT)0xFFFFFFFFFFFFull) : kdf_apply(x[0], ka));
					kv[j + 1] = (CT)(P6 ? (kdf_apply(x[1], ka) & (KT)0xFFFFFFFFFFFFull) : kdf_apply(x[1], ka));
				}
			} else {
#pragma unroll
				for (int j = 0; j < NK; ++j) {
					const u32 e = tid + BLOCK * j;
					kv[j] = e < cnt ? (CT)(P6 ? (kdf_apply(q[e], ka) & (KT)0xFFFFFFFFFFFFull) : kdf_apply(q[e], ka)) : (CT)0;
				}
			}
			{
				const u32x4 zero = {0, 0, 0, 0};
#pragma unroll
				for (int j = 0; j < PLANES; ++j)
					if (tid + BLOCK * j < (u32)NVEC)
						((u32x4 *)cell)[tid + BLOCK * j] = zero;
			}
			__syncthreads();
			auto cell_of = [&](CT v, bool valid, u32 &sh) -> u32 * {
				const u32 bin = (((u32)(v >> sh_hi) & 0xFFu) << NB2) | ((u32)(v >> sh_nx) & ((1u << NB2) - 1u));
				sh = (bin & 1u) << 4;
				return &cell[valid ? bin >> 1 : NCELLW + lane];
			};
			u32 mx = 0;
			bool handed_on = false;
			if constexpr (C::SKIP & 2) {
#pragma unroll
				for (int j = 0; j < NK; ++j)
					if (elem_of(j) < cnt)
						put_at(elem_of(j), kv[j]);
			} else {
#pragma unroll
				for (int j = 0; j < NK; ++j) {
					if ((C::VLOAD ? 2 * BLOCK * (j >> 1) : BLOCK * j) < (int)cnt) {
						u32 sh;
						u32 *a = cell_of(kv[j], elem_of(j) < cnt, sh);
						__hip_atomic_fetch_add(a, 1u << sh, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
					}
				}
				__syncthreads();
				u32x4 c[PLANES];
				u32 pk = 0, mxp = 0;
#pragma unroll
				for (int j = 0; j < PLANES; ++j) {
					c[j] = u32x4{0, 0, 0, 0};
					if (tid + BLOCK * j < (u32)NVEC)
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
				mx = (mxp & 0xFFFFu) > (mxp >> 16) ? (mxp & 0xFFFFu) : (mxp >> 16);
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
				if (mx > maxbin2) {
					if (tid == 0)
						redo[atomicAdd(&ctl->nredo, 1u)] = s;
					handed_on = true;
				}
				if (!handed_on) {
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
						if (tid + BLOCK * j < (u32)NVEC)
							((u32x4 *)cell)[tid + BLOCK * j] = x;
					}
					__syncthreads();
#pragma unroll
					for (int j = 0; j < NK; ++j) {
						if ((C::VLOAD ? 2 * BLOCK * (j >> 1) : BLOCK * j) < (int)cnt) {
							const bool valid = elem_of(j) < cnt;
							u32 sh;
							u32 *a = cell_of(kv[j], valid, sh);
							const u32 old = __hip_atomic_fetch_add(a, 1u << sh, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
							const u32 pos = (old >> sh) & 0xFFFFu;
							if constexpr (P6) {
								st_hi[valid ? at(pos) : 16 * S + lane] = (u32)((u64)kv[j] >> 16);
								st_lo[valid ? (pos & 15u) * (u32)S2 + (pos >> 4) : 16 * S2 + lane] = (unsigned short)kv[j];
							} else {
								stage[valid ? at(pos) : 16 * S + lane] = kv[j];
							}
						}
					}
				}
			}
			if (!handed_on) {
				if (tid < 32)
					put_at(cnt + tid, P6 ? (CT)0xFFFFFFFFFFFFull : (CT)~(CT)0);
				__syncthreads();
				if constexpr (!(C::SKIP & 1)) {
					const u32 npass = mx > C::MAXBIN ? 4u : 2u;
					for (u32 pass = 0; pass < npass; ++pass) {
#pragma unroll
						for (int r = 0; r < NCH; ++r) {
							const u32 ch = tid + BLOCK * r;
							const u32 off = 8 * (pass & 1);
							if (16 * ch + off < cnt) {
								CT d[16];
#pragma unroll
								for (int i = 0; i < 16; ++i)
									d[i] = (pass & 1) ? (i < 8 ? get(i + 8, ch) : get(i - 8, ch + 1)) : get(i, ch);
								if (pass == 0)
									sort16_values(d);
								else
									merge16_values(d);
#pragma unroll
								for (int i = 0; i < 16; ++i)
									if (pass & 1) {
										if (i < 8)
											put(i + 8, ch, d[i]);
										else
											put(i - 8, ch + 1, d[i]);
									} else {
										put(i, ch, d[i]);
									}
							}
						}
						__syncthreads();
					}
				}
				if constexpr (!(C::SKIP & 4)) {
					constexpr u32 CBITS = 8 * sizeof(CT);
					const KT upper = P6 ? (KT)(first >> 48 << 48) : sizeof(CT) == 8 ? (KT)0 : (KT)(first >> (CBITS & 63) << (CBITS & 63));
					KT *o = out + ls.beg;
					for (u32 i0 = 2 * tid; i0 < cnt; i0 += 2 * BLOCK) {
						KT kk[2];
						kk[0] = kdf_invert((KT)(upper | (KT)get_at(i0)), ka);
						kk[1] = kdf_invert((KT)(upper | (KT)get_at(i0 + 1)), ka);
						if (i0 + 2 <= cnt)
							store_chunk<KT, 2>(o + i0, kk);
						else
							o[i0] = kk[0];
					}
				}
			}
		}
		if constexpr (!C::LOOP)
			break;
		__syncthreads();
	}
}

}  // namespace rsx
