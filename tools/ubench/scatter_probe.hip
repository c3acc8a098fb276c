// scatter_probe: tuning harness for the scatter kernels (tile shapes, phase timeline, inputs with aligned runs).
// (The persistent / two-window / early-load experiments of round 1 -- rsx_scatter{3,4,5}_experimental.hpp -- are in the
// history up to commit 71d1354; DESIGN.md section 4 has their numbers.)
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I radix_sorting_amd/csrc -I tools/ubench tools/ubench/scatter_probe.hip -o tools/ubench/scatter_probe.bin
#include "rsx_scatter2.hpp"
#include "rsx_scatter3_two_windows.hpp"
#include "rsx_scatter4_pairs.hpp"
#include "rsx_scatter5_persistent_prefetch.hpp"
#include "rsx_scatter6_digit_waves.hpp"
#include "rsx_scatter7_rerank_windows.hpp"
#include "rsx_scatter8_pipelined.hpp"
#include "rsx_scatter11_cursor.hpp"
#include "rsx_scatter9_handoff.hpp"
#include "rsx_scatter10_one_atomic.hpp"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

using namespace rsx;

#define CK(x)                                                                         \
	do {                                                                              \
		hipError_t e_ = (x);                                                          \
		if (e_ != hipSuccess) {                                                       \
			printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
			exit(1);                                                                  \
		}                                                                             \
	} while (0)

static u32 *d_in, *d_out;
static u64 *d_hist;
static u32 *d_flag;
static Plan *d_plan;
static void *d_status;
static u64 *d_tl;
static size_t n;

static u32 g_tps = 8;
static u32 g_flags = 0;

template <typename SH, bool TL>
float run_once(u32 shift, bool timeline_dump)
{
	typedef ScatterCfg<u32, NoVal, SH> C;
	const u64 tiles = (n + C::TILE - 1) / C::TILE;
	const u64 stiles = (tiles + g_tps - 1) / g_tps;
	const size_t st_bytes = 256 + stiles * 256 * 4;
	CK(hipMemsetAsync(d_status, 0, st_bytes, 0));
	if (TL)
		CK(hipMemsetAsync(d_tl, 0, tiles * 16 * 8, 0));
	hipEvent_t e0, e1;
	CK(hipEventCreate(&e0));
	CK(hipEventCreate(&e1));
	KdfArgs<u32> ka{0, 0, 0};
	CK(hipEventRecord(e0, 0));
	hipLaunchKernelGGL((rsx_scatter_kernel<u32, NoVal, u32, SH, TL>), dim3((unsigned)stiles), dim3(C::BLOCK), 0, 0, d_in, d_out,
	                   (const NoVal *)nullptr, (NoVal *)nullptr, (u64)n, shift, d_hist + 256 * (shift / 8), g_tps,
	                   (u32 *)((char *)d_status + 256), (u32 *)d_status, ka, g_flags, d_tl);
	CK(hipGetLastError());
	CK(hipEventRecord(e1, 0));
	CK(hipEventSynchronize(e1));
	float ms;
	CK(hipEventElapsedTime(&ms, e0, e1));
	if (TL && timeline_dump) {
		std::vector<u64> tl(tiles * 16);
		CK(hipMemcpy(tl.data(), d_tl, tiles * 16 * 8, hipMemcpyDeviceToHost));
		// stamps per tile: 2 loaded, 3 ranked, 4 sync1, 6 counts+sync2, 7 bases+sync3, 8 staged, 9 sync4, 10 written
		// per super-tile (row of its first tile): 0 start, 1 phase A + look-back done, 12 depth
		const int seq[] = {2, 3, 4, 6, 7, 8, 9, 10};
		const char *names[] = {"rank", "sync1", "counts+sync2", "bases+sync3", "stage", "sync4", "writeout"};
		double sum[7] = {0}, lookb = 0, phaseA = 0, depth = 0, maxdepth = 0, loadw = 0, span = 0;
		u64 nst = 0;
		for (u64 t = 0; t < tiles; ++t) {
			const u64 *r = &tl[t * 16];
			for (int k = 0; k < 7; ++k)
				sum[k] += (double)(r[seq[k + 1]] - r[seq[k]]);
			if (t % g_tps == 0) {
				phaseA += (double)(r[13] - r[0]);
				lookb += (double)(r[1] - r[13]);
				depth += r[12];
				maxdepth = std::max(maxdepth, (double)r[12]);
				loadw += (double)(r[2] - r[1]);
				const u64 last = std::min(tiles - 1, t + g_tps - 1);
				span += (double)(tl[last * 16 + 10] - r[0]);
				++nst;
			} else {
				loadw += (double)(r[2] - tl[(t - 1) * 16 + 10]);
			}
		}
		printf("  per super-tile: phase A %9.0f ticks, look-back %9.0f ticks, depth avg %.2f max %.0f, lifetime %9.0f ticks\n",
		       phaseA / nst, lookb / nst, depth / nst, maxdepth, span / nst);
		printf("  per tile: load wait %7.0f", loadw / tiles);
		for (int k = 0; k < 7; ++k)
			printf(" | %s %6.0f", names[k], sum[k] / tiles);
		printf("\n");
	}
	return ms;
}

template <typename C, bool TL, bool HOTV = false>
float run2_once(u32 shift, bool dump, u32 tps)
{
	const u64 tiles = (n + C::TILE - 1) / C::TILE;
	const u64 stiles = (tiles + tps - 1) / tps;
	CK(hipMemsetAsync(d_status, 0, 256 + stiles * 256 * 4, 0));
	if (TL)
		CK(hipMemsetAsync(d_tl, 0, stiles * 16 * 8, 0));
	hipEvent_t e0, e1;
	CK(hipEventCreate(&e0));
	CK(hipEventCreate(&e1));
	KdfArgs<u32> ka{0, 0, 0};
	CK(hipEventRecord(e0, 0));
	hipLaunchKernelGGL((rsx_scatter2_kernel<u32, NoVal, u32, C, TL, DIG_PLAIN, HOTV>), dim3((unsigned)stiles), dim3(C::BLOCK), 0, 0, d_in,
	                   d_out, (const NoVal *)nullptr, (NoVal *)nullptr, (u64)n, shift, d_hist + 256 * (shift / 8), tps,
	                   (u32 *)((char *)d_status + 256), (u32 *)d_status, ka, g_flags | ((shift / 8) << SCATTER_COL_SHIFT),
	                   d_tl, (const Plan *)nullptr, 0u, 0u, (const u32 *)(d_flag + 32));
	CK(hipGetLastError());
	CK(hipEventRecord(e1, 0));
	CK(hipEventSynchronize(e1));
	float ms;
	CK(hipEventElapsedTime(&ms, e0, e1));
	if (TL && dump) {
		std::vector<u64> tl(stiles * 16);
		CK(hipMemcpy(tl.data(), d_tl, stiles * 16 * 8, hipMemcpyDeviceToHost));
		double a = 0, lb = 0, lay = 0, st[4] = {0}, wo[4] = {0}, depth = 0, life = 0;
		for (u64 s = 0; s < stiles; ++s) {
			const u64 *r = &tl[s * 16];
			a += (double)(r[1] - r[0]);
			lay += (double)(r[2] - r[1]);
			lb += (double)(r[3] - r[2]);
			u64 prev = r[2];   // staging starts after the layout; the digit threads resolve the chain first
			for (u32 t = 0; t < tps && t < 4; ++t) {
				st[t] += (double)(r[4 + 2 * t] - prev);
				wo[t] += (double)(r[5 + 2 * t] - r[4 + 2 * t]);
				prev = r[5 + 2 * t];
			}
			life += (double)(prev - r[0]);
			depth += r[12];
		}
		printf("  per super-tile: phase A %8.0f | layout %6.0f | chain %7.0f (depth %.1f) inside |", a / stiles, lay / stiles, lb / stiles,
		       depth / stiles);
		for (u32 t = 0; t < tps && t < 4; ++t)
			printf(" stage%u %7.0f write%u %7.0f |", t, st[t] / stiles, t, wo[t] / stiles);
		printf(" lifetime %8.0f\n", life / stiles);
	}
	return ms;
}

template <typename C, bool TL>
float run3_once(u32 shift, bool dump)
{
	const u64 tiles = (n + C::TILE - 1) / C::TILE;
	CK(hipMemsetAsync(d_status, 0, 256 + tiles * 256 * 4, 0));
	if (TL)
		CK(hipMemsetAsync(d_tl, 0, tiles * 16 * 8, 0));
	hipEvent_t e0, e1;
	CK(hipEventCreate(&e0));
	CK(hipEventCreate(&e1));
	KdfArgs<u32> ka{0, 0, 0};
	CK(hipEventRecord(e0, 0));
	hipLaunchKernelGGL((rsx_scatter3_kernel<u32, u32, C, TL, DIG_PLAIN>), dim3((unsigned)tiles), dim3(C::BLOCK), 0, 0, d_in, d_out, (u64)n,
	                   shift, d_hist + 256 * (shift / 8), (u32 *)((char *)d_status + 256), (u32 *)d_status, ka, g_flags, d_tl,
	                   (const Plan *)nullptr, 0u);
	CK(hipGetLastError());
	CK(hipEventRecord(e1, 0));
	CK(hipEventSynchronize(e1));
	float ms;
	CK(hipEventElapsedTime(&ms, e0, e1));
	CK(hipEventDestroy(e0));
	CK(hipEventDestroy(e1));
	if (TL && dump) {
		std::vector<u64> tl(tiles * 16);
		CK(hipMemcpy(tl.data(), d_tl, tiles * 16 * 8, hipMemcpyDeviceToHost));
		double a = 0, lay = 0, rk = 0, ch = 0, w0 = 0, s1 = 0, w1 = 0, depth = 0, life = 0;
		for (u64 t = 0; t < tiles; ++t) {
			const u64 *r = &tl[t * 16];
			a += (double)(r[1] - r[0]);
			lay += (double)(r[2] - r[1]);
			ch += (double)(r[3] - r[2]);
			rk += (double)(r[4] - r[2]);
			w0 += (double)(r[5] - r[4]);
			s1 += (double)(r[6] - r[5]);
			w1 += (double)(r[7] - r[6]);
			life += (double)(r[7] - r[0]);
			depth += r[12];
		}
		printf("  per tile: load+count %7.0f | layout %6.0f | rank+stage0 (chain inside) %7.0f (digit thread 0: chain done after %7.0f, depth %.1f) | "
		       "write0 %6.0f | stage1 %6.0f | write1 %6.0f | lifetime %7.0f\n",
		       a / tiles, lay / tiles, rk / tiles, ch / tiles, depth / tiles, w0 / tiles, s1 / tiles, w1 / tiles, life / tiles);
	}
	return ms;
}

template <typename C>
void bench3(const char *name)
{
	run3_once<C, false>(0, false);
	float best = 1e9, sum = 0;
	const int reps = 5;
	for (int i = 0; i < reps; ++i) {
		float ms = run3_once<C, false>(8 * (i % 4), false);
		best = std::min(best, ms);
		sum += ms;
	}
	printf("%-14s tile %6d lds %6zu B: avg %.3f ms best %.3f ms  -> %.0f GB/s (algorithmic 8 B/key)\n", name, C::TILE,
	       sizeof(Sc3Smem<u32, u32, C>), sum / reps, best, n * 8.0 / (best * 1e-3) / 1e9);
	run3_once<C, true>(0, true);
	// same output as the reference kernel of this probe (rsx_scatter2_kernel)?
	std::vector<u32> a(1 << 22), b(1 << 22);
	run3_once<C, false>(8, false);
	CK(hipMemcpy(a.data(), d_out + (n / 2 - (1 << 21)), a.size() * 4, hipMemcpyDeviceToHost));
	run2_once<Sc2Cfg<u32, NoVal>, false, false>(8, false, 1);
	CK(hipMemcpy(b.data(), d_out + (n / 2 - (1 << 21)), b.size() * 4, hipMemcpyDeviceToHost));
	printf("  %s\n", a == b ? "output identical to rsx_scatter2_kernel's (2^22 keys around the middle, column 1)" : "OUTPUT DIFFERS from rsx_scatter2_kernel's");
	g_flags = SCATTER_DBG_NOSTORE;
	printf("  without global stores: %.3f ms\n", run3_once<C, true>(0, false));
	g_flags = 0;
}

template <typename C, bool TL>
float run4_once(u32 shift, bool dump)
{
	const u64 npairs = n / (2 * (u64)C::TILE);
	CK(hipMemsetAsync(d_status, 0, 256 + npairs * 2 * 256 * 4, 0));
	if (TL)
		CK(hipMemsetAsync(d_tl, 0, npairs * 16 * 8, 0));
	hipEvent_t e0, e1;
	CK(hipEventCreate(&e0));
	CK(hipEventCreate(&e1));
	KdfArgs<u32> ka{0, 0, 0};
	CK(hipEventRecord(e0, 0));
	hipLaunchKernelGGL((rsx_scatter4_kernel<u32, u32, C, TL, DIG_PLAIN>), dim3((unsigned)npairs), dim3(C::BLOCK), 0, 0, d_in, d_out, npairs,
	                   shift, d_hist + 256 * (shift / 8), (u32 *)((char *)d_status + 256), (u32 *)d_status, ka, g_flags, d_tl,
	                   (const Plan *)nullptr, 0u);
	CK(hipGetLastError());
	CK(hipEventRecord(e1, 0));
	CK(hipEventSynchronize(e1));
	float ms;
	CK(hipEventElapsedTime(&ms, e0, e1));
	CK(hipEventDestroy(e0));
	CK(hipEventDestroy(e1));
	if (TL && dump) {
		std::vector<u64> tl(npairs * 16);
		CK(hipMemcpy(tl.data(), d_tl, npairs * 16 * 8, hipMemcpyDeviceToHost));
		double a = 0, lay = 0, ch = 0, sa = 0, wa = 0, lb = 0, sb = 0, wb = 0, depth = 0, life = 0;
		for (u64 t = 0; t < npairs; ++t) {
			const u64 *r = &tl[t * 16];
			a += (double)(r[1] - r[0]);
			lay += (double)(r[2] - r[1]);
			ch += (double)(r[3] - r[2]);
			sa += (double)(r[4] - r[2]);
			wa += (double)(r[5] - r[4]);
			lb += (double)(r[6] - r[5]);
			sb += (double)(r[7] - r[6]);
			wb += (double)(r[8] - r[7]);
			life += (double)(r[8] - r[0]);
			depth += r[12];
		}
		printf("  per pair: load+count %7.0f | layout A %6.0f | stage A %7.0f (chain %6.0f, depth %.1f tiles, inside) | write A %6.0f | layout B %6.0f | "
		       "stage B %6.0f | write B %6.0f | lifetime %7.0f (%.0f per tile)\n",
		       a / npairs, lay / npairs, sa / npairs, ch / npairs, depth / npairs, wa / npairs, lb / npairs, sb / npairs, wb / npairs, life / npairs,
		       life / npairs / 2);
	}
	return ms;
}

template <typename C>
void bench4(const char *name)
{
	run4_once<C, false>(0, false);
	float best = 1e9, sum = 0;
	const int reps = 5;
	for (int i = 0; i < reps; ++i) {
		float ms = run4_once<C, false>(8 * (i % 4), false);
		best = std::min(best, ms);
		sum += ms;
	}
	printf("%-14s tile %6d x 2, lds %6zu B: avg %.3f ms best %.3f ms  -> %.0f GB/s (algorithmic 8 B/key)\n", name, C::TILE,
	       sizeof(Sc4Smem<u32, u32, C>), sum / reps, best, n * 8.0 / (best * 1e-3) / 1e9);
	run4_once<C, true>(0, true);
	// same output as the reference kernel of this probe (rsx_scatter2_kernel)?  the whole array, column 1
	std::vector<u32> a(n), b(n);
	run4_once<C, false>(8, false);
	CK(hipMemcpy(a.data(), d_out, n * 4, hipMemcpyDeviceToHost));
	run2_once<Sc2Cfg<u32, NoVal>, false, false>(8, false, 1);
	CK(hipMemcpy(b.data(), d_out, n * 4, hipMemcpyDeviceToHost));
	printf("  %s\n", a == b ? "output identical to rsx_scatter2_kernel's (the whole array, column 1)" : "OUTPUT DIFFERS from rsx_scatter2_kernel's");
	g_flags = SCATTER_DBG_NOSTORE;
	printf("  without global stores: %.3f ms\n", run4_once<C, true>(0, false));
	g_flags = 0;
}

template <typename C, bool TL>
float run5_once(u32 shift, bool dump, u32 grid)
{
	const u32 ntiles = (u32)(n / C::TILE);
	CK(hipMemsetAsync(d_status, 0, 256 + (size_t)ntiles * 256 * 4, 0));
	if (TL)
		CK(hipMemsetAsync(d_tl, 0, (size_t)ntiles * 16 * 8, 0));
	hipEvent_t e0, e1;
	CK(hipEventCreate(&e0));
	CK(hipEventCreate(&e1));
	KdfArgs<u32> ka{0, 0, 0};
	CK(hipEventRecord(e0, 0));
	hipLaunchKernelGGL((rsx_scatter5_kernel<u32, u32, C, TL, DIG_PLAIN>), dim3(grid), dim3(C::BLOCK), 0, 0, d_in, d_out, ntiles, shift,
	                   d_hist + 256 * (shift / 8), (u32 *)((char *)d_status + 256), (u32 *)d_status, ka, g_flags, d_tl, (const Plan *)nullptr,
	                   0u);
	CK(hipGetLastError());
	CK(hipEventRecord(e1, 0));
	CK(hipEventSynchronize(e1));
	float ms;
	CK(hipEventElapsedTime(&ms, e0, e1));
	CK(hipEventDestroy(e0));
	CK(hipEventDestroy(e1));
	if (TL && dump) {
		std::vector<u64> tl(ntiles * 16);
		CK(hipMemcpy(tl.data(), d_tl, (size_t)ntiles * 16 * 8, hipMemcpyDeviceToHost));
		double a = 0, lay = 0, ch = 0, st = 0, wo = 0, depth = 0, life = 0;
		for (u64 t = 0; t < ntiles; ++t) {
			const u64 *r = &tl[t * 16];
			a += (double)(r[1] - r[0]);
			lay += (double)(r[2] - r[1]);
			ch += (double)(r[3] - r[2]);
			st += (double)(r[4] - r[2]);
			wo += (double)(r[5] - r[4]);
			life += (double)(r[5] - r[0]);
			depth += r[12];
		}
		printf("  per tile: wait + count %7.0f | layout %6.0f | stage %7.0f (chain %6.0f, depth %.1f, inside) | write-out %6.0f | lifetime %7.0f\n",
		       a / ntiles, lay / ntiles, st / ntiles, ch / ntiles, depth / ntiles, wo / ntiles, life / ntiles);
	}
	return ms;
}

template <typename C>
void bench5(const char *name, u32 grid)
{
	run5_once<C, false>(0, false, grid);
	float best = 1e9, sum = 0;
	const int reps = 5;
	for (int i = 0; i < reps; ++i) {
		float ms = run5_once<C, false>(8 * (i % 4), false, grid);
		best = std::min(best, ms);
		sum += ms;
	}
	printf("%-14s grid %u tile %6d, lds %6zu B: avg %.3f ms best %.3f ms  -> %.0f GB/s (algorithmic 8 B/key)\n", name, grid, C::TILE,
	       sizeof(Sc5Smem<u32, u32, C>), sum / reps, best, n * 8.0 / (best * 1e-3) / 1e9);
	run5_once<C, true>(0, true, grid);
	std::vector<u32> a(n), b(n);
	run5_once<C, false>(8, false, grid);
	CK(hipMemcpy(a.data(), d_out, n * 4, hipMemcpyDeviceToHost));
	run2_once<Sc2Cfg<u32, NoVal>, false, false>(8, false, 1);
	CK(hipMemcpy(b.data(), d_out, n * 4, hipMemcpyDeviceToHost));
	printf("  %s\n", a == b ? "output identical to rsx_scatter2_kernel's (the whole array, column 1)" : "OUTPUT DIFFERS from rsx_scatter2_kernel's");
	g_flags = SCATTER_DBG_NOSTORE;
	printf("  without global stores: %.3f ms\n", run5_once<C, true>(0, false, grid));
	g_flags = 0;
}

template <typename C, bool TL>
float run6_once(u32 shift, bool dump)
{
	const u64 tiles = (n + C::TILE - 1) / C::TILE;
	CK(hipMemsetAsync(d_status, 0, 256 + tiles * 256 * 4, 0));
	if (TL)
		CK(hipMemsetAsync(d_tl, 0, tiles * 16 * 8, 0));
	hipEvent_t e0, e1;
	CK(hipEventCreate(&e0));
	CK(hipEventCreate(&e1));
	KdfArgs<u32> ka{0, 0, 0};
	CK(hipEventRecord(e0, 0));
	hipLaunchKernelGGL((rsx_scatter6_kernel<u32, u32, C, TL, DIG_PLAIN>), dim3((unsigned)tiles), dim3(C::BLOCK), 0, 0, d_in, d_out, (u64)n,
	                   shift, d_hist + 256 * (shift / 8), (u32 *)((char *)d_status + 256), (u32 *)d_status, ka, g_flags, d_tl,
	                   (const Plan *)nullptr, 0u);
	CK(hipGetLastError());
	CK(hipEventRecord(e1, 0));
	CK(hipEventSynchronize(e1));
	float ms;
	CK(hipEventElapsedTime(&ms, e0, e1));
	CK(hipEventDestroy(e0));
	CK(hipEventDestroy(e1));
	if (TL && dump) {
		std::vector<u64> tl(tiles * 16);
		CK(hipMemcpy(tl.data(), d_tl, tiles * 16 * 8, hipMemcpyDeviceToHost));
		double a = 0, lay = 0, ch = 0, st = 0, kw = 0, wo = 0, depth = 0, life = 0;
		const u64 nt = tiles - 1;   // (whole tiles)
		for (u64 t = 0; t < nt; ++t) {
			const u64 *r = &tl[t * 16];
			a += (double)(r[1] - r[0]);
			lay += (double)(r[2] - r[1]);
			ch += (double)(r[3] - r[2]);
			kw += (double)(r[6] - r[2]);
			st += (double)(r[4] - r[2]);
			wo += (double)(r[5] - r[4]);
			life += (double)(r[5] - r[0]);
			depth += r[12];
		}
		printf("  per tile: load+count %7.0f | layout %6.0f | stage phase %7.0f (chain %6.0f, depth %.1f; key wave 0 staged after %6.0f) | write-out %6.0f | "
		       "lifetime %7.0f (%.0f per 32 Ki keys)\n",
		       a / nt, lay / nt, st / nt, ch / nt, depth / nt, kw / nt, wo / nt, life / nt, life / nt * 32768.0 / C::TILE);
	}
	return ms;
}

template <typename C>
void bench6(const char *name)
{
	run6_once<C, false>(0, false);
	float best = 1e9, sum = 0;
	const int reps = 5;
	for (int i = 0; i < reps; ++i) {
		float ms = run6_once<C, false>(8 * (i % 4), false);
		best = std::min(best, ms);
		sum += ms;
	}
	printf("%-14s tile %6d, lds %6zu B: avg %.3f ms best %.3f ms  -> %.0f GB/s (algorithmic 8 B/key)\n", name, C::TILE,
	       sizeof(Sc6Smem<u32, u32, C>), sum / reps, best, n * 8.0 / (best * 1e-3) / 1e9);
	run6_once<C, true>(0, true);
	std::vector<u32> a(n), b(n);
	run6_once<C, false>(8, false);
	CK(hipMemcpy(a.data(), d_out, n * 4, hipMemcpyDeviceToHost));
	run2_once<Sc2Cfg<u32, NoVal>, false, false>(8, false, 1);
	CK(hipMemcpy(b.data(), d_out, n * 4, hipMemcpyDeviceToHost));
	printf("  %s\n", a == b ? "output identical to rsx_scatter2_kernel's (the whole array, column 1)" : "OUTPUT DIFFERS from rsx_scatter2_kernel's");
	g_flags = SCATTER_DBG_NOSTORE;
	printf("  without global stores: %.3f ms\n", run6_once<C, true>(0, false));
	g_flags = 0;
}

template <typename C, bool TL>
float run7_once(u32 shift, bool dump)
{
	const u64 tiles = n / C::TILE;
	CK(hipMemsetAsync(d_status, 0, 256 + tiles * 256 * 4, 0));
	if (TL)
		CK(hipMemsetAsync(d_tl, 0, tiles * 16 * 8, 0));
	hipEvent_t e0, e1;
	CK(hipEventCreate(&e0));
	CK(hipEventCreate(&e1));
	KdfArgs<u32> ka{0, 0, 0};
	CK(hipEventRecord(e0, 0));
	hipLaunchKernelGGL((rsx_scatter7_kernel<u32, u32, C, TL, DIG_PLAIN>), dim3((unsigned)tiles), dim3(C::BLOCK), 0, 0, d_in, d_out, (u64)n,
	                   shift, d_hist + 256 * (shift / 8), (u32 *)((char *)d_status + 256), (u32 *)d_status, ka, g_flags, d_tl);
	CK(hipGetLastError());
	CK(hipEventRecord(e1, 0));
	CK(hipEventSynchronize(e1));
	float ms;
	CK(hipEventElapsedTime(&ms, e0, e1));
	CK(hipEventDestroy(e0));
	CK(hipEventDestroy(e1));
	if (TL && dump) {
		std::vector<u64> tl(tiles * 16);
		CK(hipMemcpy(tl.data(), d_tl, tiles * 16 * 8, hipMemcpyDeviceToHost));
		double a = 0, lay = 0, ch = 0, s0 = 0, w0 = 0, s1 = 0, w1 = 0, depth = 0, life = 0;
		for (u64 t = 0; t < tiles; ++t) {
			const u64 *r = &tl[t * 16];
			a += (double)(r[1] - r[0]);
			lay += (double)(r[2] - r[1]);
			ch += (double)(r[3] - r[2]);
			s0 += (double)(r[4] - r[2]);
			w0 += (double)(r[5] - r[4]);
			s1 += (double)(r[6] - r[5]);
			w1 += (double)(r[7] - r[6]);
			life += (double)(r[7] - r[0]);
			depth += r[12];
		}
		printf("  per tile: load+count %7.0f | layout %6.0f | rank+stage0 %7.0f (chain %6.0f, depth %.1f, inside) | write0 %6.0f | rank+stage1 %6.0f | "
		       "write1 %6.0f | lifetime %7.0f\n",
		       a / tiles, lay / tiles, s0 / tiles, ch / tiles, depth / tiles, w0 / tiles, s1 / tiles, w1 / tiles, life / tiles);
	}
	return ms;
}

template <typename C>
void bench7(const char *name)
{
	run7_once<C, false>(0, false);
	float best = 1e9, sum = 0;
	const int reps = 5;
	for (int i = 0; i < reps; ++i) {
		float ms = run7_once<C, false>(8 * (i % 4), false);
		best = std::min(best, ms);
		sum += ms;
	}
	printf("%-14s tile %6d lds %6zu B: avg %.3f ms best %.3f ms  -> %.0f GB/s (algorithmic 8 B/key)\n", name, C::TILE,
	       sizeof(Sc7Smem<u32, u32, C>), sum / reps, best, n * 8.0 / (best * 1e-3) / 1e9);
	run7_once<C, true>(0, true);
	std::vector<u32> a(n), b(n);
	run7_once<C, false>(8, false);
	CK(hipMemcpy(a.data(), d_out, n * 4, hipMemcpyDeviceToHost));
	run2_once<Sc2Cfg<u32, NoVal>, false, false>(8, false, 1);
	CK(hipMemcpy(b.data(), d_out, n * 4, hipMemcpyDeviceToHost));
	printf("  %s\n", a == b ? "output identical to rsx_scatter2_kernel's (the whole array, column 1)" : "OUTPUT DIFFERS from rsx_scatter2_kernel's");
	g_flags = SCATTER_DBG_NOSTORE;
	printf("  without global stores: %.3f ms\n", run7_once<C, true>(0, false));
	g_flags = 0;
}

// EXPERIMENT (round 2): a helper kernel on a second stream reads the input LEAD tiles ahead of the scatter kernel's ticket
// front (so that the lines are in the memory-side cache, and in one XCD's L2, when a scatter workgroup asks for them).
// Its workgroups use no LDS, so they share CUs with the scatter workgroups.
__global__ __launch_bounds__(256) void read_ahead_kernel(const uint4 *__restrict__ in, u64 nchunks, const u32 *ticket, u32 chunks_per_tile,
                                                         u32 lead_tiles, u32 ntiles, u32 *sink)
{
	const u32 G = gridDim.x;
	u64 c = blockIdx.x;
	u32 acc = 0;
	for (;;) {
		const u32 tk = __hip_atomic_load(ticket, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		if (tk >= ntiles)
			break;
		const u64 lo = (u64)tk * chunks_per_tile;
		u64 hi = (u64)(tk + lead_tiles) * chunks_per_tile;
		if (hi > nchunks)
			hi = nchunks;
		if (c < lo)
			c += (lo - c + G - 1) / G * G;
		if (c >= nchunks)
			break;
		if (c >= hi) {
			__builtin_amdgcn_s_sleep(8);
			continue;
		}
		uint4 v[8];
#pragma unroll
		for (int j = 0; j < 8; ++j) {
			const u64 cc = c + (u64)j * G < nchunks ? c + (u64)j * G : c;
			v[j] = in[cc * 256 + threadIdx.x];
		}
#pragma unroll
		for (int j = 0; j < 8; ++j)
			acc ^= v[j].x;
		c += 8ull * G;
	}
	if (acc == 0x12345679u)
		*sink = acc;
}

static hipStream_t g_side;

float run2_read_ahead(u32 shift, u32 lead, u32 wgs)
{
	typedef Sc2Cfg<u32, NoVal> C;
	const u64 tiles = (n + C::TILE - 1) / C::TILE;
	if (!g_side)
		CK(hipStreamCreateWithFlags(&g_side, hipStreamNonBlocking));
	CK(hipMemsetAsync(d_status, 0, 256 + tiles * 256 * 4, 0));
	hipEvent_t e0, e1, e2;
	CK(hipEventCreate(&e0));
	CK(hipEventCreate(&e1));
	CK(hipEventCreate(&e2));
	KdfArgs<u32> ka{0, 0, 0};
	CK(hipEventRecord(e0, 0));
	CK(hipStreamWaitEvent(g_side, e0, 0));
	hipLaunchKernelGGL((rsx_scatter2_kernel<u32, NoVal, u32, C, false, DIG_PLAIN, false>), dim3((unsigned)tiles), dim3(C::BLOCK), 0, 0, d_in,
	                   d_out, (const NoVal *)nullptr, (NoVal *)nullptr, (u64)n, shift, d_hist + 256 * (shift / 8), 1u,
	                   (u32 *)((char *)d_status + 256), (u32 *)d_status, ka, g_flags | ((shift / 8) << SCATTER_COL_SHIFT), d_tl,
	                   (const Plan *)nullptr, 0u, 0u, (const u32 *)(d_flag + 32));
	hipLaunchKernelGGL(read_ahead_kernel, dim3(wgs), dim3(256), 0, g_side, (const uint4 *)d_in, (u64)(n * 4 / 4096), (const u32 *)d_status,
	                   (u32)(C::TILE * 4 / 4096), lead, (u32)tiles, d_flag + 60);
	CK(hipGetLastError());
	CK(hipEventRecord(e2, g_side));
	CK(hipStreamWaitEvent(0, e2, 0));
	CK(hipEventRecord(e1, 0));
	CK(hipEventSynchronize(e1));
	float ms;
	CK(hipEventElapsedTime(&ms, e0, e1));
	CK(hipEventDestroy(e0));
	CK(hipEventDestroy(e1));
	CK(hipEventDestroy(e2));
	return ms;
}

void bench_read_ahead()
{
	const u32 leads[] = {64, 256, 1024};
	const u32 wgss[] = {32, 128, 512};
	for (u32 lead : leads)
		for (u32 wgs : wgss) {
			run2_read_ahead(0, lead, wgs);
			float best = 1e9, sum = 0;
			for (int i = 0; i < 5; ++i) {
				const float ms = run2_read_ahead(8 * (i % 4), lead, wgs);
				best = std::min(best, ms);
				sum += ms;
			}
			printf("v2 + read-ahead helper: lead %4u tiles, %3u workgroups: avg %.3f ms best %.3f ms\n", lead, wgs, sum / 5, best);
		}
	// same output?
	std::vector<u32> a(n), b(n);
	run2_read_ahead(8, 256, 128);
	CK(hipMemcpy(a.data(), d_out, n * 4, hipMemcpyDeviceToHost));
	run2_once<Sc2Cfg<u32, NoVal>, false, false>(8, false, 1);
	CK(hipMemcpy(b.data(), d_out, n * 4, hipMemcpyDeviceToHost));
	printf("  %s\n", a == b ? "output identical with and without the helper" : "OUTPUT DIFFERS with the helper");
}

// v11: the pipelined persistent pass with atomic cursors instead of the chain (rsx_scatter11_cursor.hpp); output partitioned by
// digit but unstable across tiles: checked for digit order and for the keys' sum / xor
__global__ void partition_check_kernel(const u32 *a, u64 n, u32 shift, u64 *out)
{
	u64 bad = 0, sum = 0, x = 0;
	for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) {
		if (i + 1 < n && ((a[i] >> shift) & 0xFFu) > ((a[i + 1] >> shift) & 0xFFu))
			++bad;
		sum += a[i];
		x ^= (u64)a[i] * 0x9E3779B97F4A7C15ull;
	}
	atomicAdd((unsigned long long *)&out[0], bad);
	atomicAdd((unsigned long long *)&out[1], sum);
	atomicXor((unsigned long long *)&out[2], x);
}
static u32 *d_cursor;
template <typename C, bool TL>
float run11_once(u32 shift, bool dump, u32 grid)
{
	const u32 ntiles = (u32)(n / C::TILE);
	if (!d_cursor)
		CK(hipMalloc(&d_cursor, 1024));
	CK(hipMemsetAsync(d_cursor, 0, 1024, 0));
	CK(hipMemsetAsync(d_status, 0, 256, 0));
	if (TL)
		CK(hipMemsetAsync(d_tl, 0, (size_t)ntiles * 16 * 8, 0));
	hipEvent_t e0, e1;
	CK(hipEventCreate(&e0));
	CK(hipEventCreate(&e1));
	KdfArgs<u32> ka{0, 0, 0};
	CK(hipEventRecord(e0, 0));
	hipLaunchKernelGGL((rsx_scatter11_kernel<u32, u32, C, TL, DIG_PLAIN>), dim3(grid), dim3(C::BLOCK), 0, 0, d_in, d_out, ntiles, shift,
	                   d_hist + 256 * (shift / 8), d_cursor, (u32 *)d_status, ka, g_flags, d_tl);
	CK(hipGetLastError());
	CK(hipEventRecord(e1, 0));
	CK(hipEventSynchronize(e1));
	float ms;
	CK(hipEventElapsedTime(&ms, e0, e1));
	CK(hipEventDestroy(e0));
	CK(hipEventDestroy(e1));
	if (TL && dump) {
		std::vector<u64> tl(ntiles * 16);
		CK(hipMemcpy(tl.data(), d_tl, (size_t)ntiles * 16 * 8, hipMemcpyDeviceToHost));
		double lay = 0, st = 0, bar = 0, wr = 0, life = 0;
		u64 cnt = 0;
		for (u64 t = 0; t < ntiles; ++t) {
			const u64 *r = &tl[t * 16];
			lay += (double)(r[1] - r[0]);
			st += (double)(r[2] - r[1]);
			bar += (double)(r[3] - r[1]);
			wr += (double)(r[4] - r[3]);
			life += (double)(r[4] - r[0]);
			++cnt;
		}
		printf("  per tile: layout + cursor atomic %6.0f | prefetch issue + stage (wave 0) %6.0f, barrier at %6.0f | write-out + rank next %6.0f | period %7.0f\n",
		       lay / cnt, st / cnt, bar / cnt, wr / cnt, life / cnt);
	}
	return ms;
}

template <typename C>
void bench11(const char *name, u32 grid)
{
	run11_once<C, false>(0, false, grid);
	float best = 1e9, sum = 0;
	const int reps = 5;
	for (int i = 0; i < reps; ++i) {
		float ms = run11_once<C, false>(8 * (i % 4), false, grid);
		best = std::min(best, ms);
		sum += ms;
	}
	printf("%-26s grid %u tile %6d: avg %.3f ms best %.3f ms  -> %.0f GB/s (algorithmic 8 B/key)\n", name, grid, C::TILE, sum / reps, best,
	       n * 8.0 / (best * 1e-3) / 1e9);
	run11_once<C, true>(0, true, grid);
	for (u32 shift = 0; shift < 32; shift += 8) {
		run11_once<C, false>(shift, false, grid);
		u64 *d_chk, chk[6];
		CK(hipMalloc(&d_chk, 48));
		CK(hipMemset(d_chk, 0, 48));
		hipLaunchKernelGGL(partition_check_kernel, dim3(2048), dim3(256), 0, 0, (const u32 *)d_out, (u64)n, shift, d_chk);
		hipLaunchKernelGGL(partition_check_kernel, dim3(2048), dim3(256), 0, 0, (const u32 *)d_in, (u64)n, shift, d_chk + 3);
		CK(hipMemcpy(chk, d_chk, 48, hipMemcpyDeviceToHost));
		printf("  column %u: digit-order violations %llu, keys %s\n", shift / 8, (unsigned long long)chk[0],
		       chk[1] == chk[4] && chk[2] == chk[5] ? "preserved" : "CHANGED");
		CK(hipFree(d_chk));
	}
	const u32 keepf = g_flags;
	g_flags = SCATTER_DBG_NOSTORE;
	printf("  without global stores: %.3f ms\n", run11_once<C, true>(0, false, grid));
	g_flags = keepf;
}

template <typename C, bool TL>
float run8_once(u32 shift, bool dump, u32 grid)
{
	const u32 ntiles = (u32)(n / C::TILE);
	CK(hipMemsetAsync(d_status, 0, 256 + (size_t)ntiles * 256 * 4, 0));
	if (TL)
		CK(hipMemsetAsync(d_tl, 0, (size_t)ntiles * 16 * 8, 0));
	hipEvent_t e0, e1;
	CK(hipEventCreate(&e0));
	CK(hipEventCreate(&e1));
	KdfArgs<u32> ka{0, 0, 0};
	CK(hipEventRecord(e0, 0));
	hipLaunchKernelGGL((rsx_scatter8_kernel<u32, u32, C, TL, DIG_PLAIN>), dim3(grid), dim3(C::BLOCK), 0, 0, d_in, d_out, ntiles, shift,
	                   d_hist + 256 * (shift / 8), (u32 *)((char *)d_status + 256), (u32 *)d_status, ka, g_flags, d_tl);
	CK(hipGetLastError());
	CK(hipEventRecord(e1, 0));
	CK(hipEventSynchronize(e1));
	float ms;
	CK(hipEventElapsedTime(&ms, e0, e1));
	CK(hipEventDestroy(e0));
	CK(hipEventDestroy(e1));
	if (TL && dump) {
		std::vector<u64> tl(ntiles * 16);
		CK(hipMemcpy(tl.data(), d_tl, (size_t)ntiles * 16 * 8, hipMemcpyDeviceToHost));
		double lay = 0, st = 0, bar = 0, wr = 0, depth = 0, steps = 0, life = 0;
		u64 cnt = 0;
		for (u64 t = 0; t < ntiles; ++t) {
			const u64 *r = &tl[t * 16];
			lay += (double)(r[1] - r[0]);
			st += (double)(r[2] - r[1]);
			bar += (double)(r[3] - r[1]);
			wr += (double)(r[4] - r[3]);
			life += (double)(r[4] - r[0]);
			depth += r[12];
			steps += r[13];
			++cnt;
		}
		printf("  per tile: layout + chain %6.0f (depth %.1f, %.2f steps) | prefetch issue + stage (wave 0) %6.0f, barrier at %6.0f | write-out + rank next %6.0f | "
		       "period %7.0f\n",
		       lay / cnt, depth / cnt, steps / cnt, st / cnt, bar / cnt, wr / cnt, life / cnt);
	}
	return ms;
}

template <typename C>
void bench8(const char *name, u32 grid)
{
	run8_once<C, false>(0, false, grid);
	float best = 1e9, sum = 0;
	const int reps = 5;
	for (int i = 0; i < reps; ++i) {
		float ms = run8_once<C, false>(8 * (i % 4), false, grid);
		best = std::min(best, ms);
		sum += ms;
	}
	printf("%-14s grid %u tile %6d, lds %6zu B: avg %.3f ms best %.3f ms  -> %.0f GB/s (algorithmic 8 B/key)\n", name, grid, C::TILE,
	       sizeof(Sc8Smem<u32, u32, C>), sum / reps, best, n * 8.0 / (best * 1e-3) / 1e9);
	run8_once<C, true>(0, true, grid);
	std::vector<u32> a(n), b(n);
	run8_once<C, false>(8, false, grid);
	CK(hipMemcpy(a.data(), d_out, n * 4, hipMemcpyDeviceToHost));
	const u32 keepf = g_flags;
	g_flags = 0;
	run2_once<Sc2Cfg<u32, NoVal>, false, false>(8, false, 1);
	CK(hipMemcpy(b.data(), d_out, n * 4, hipMemcpyDeviceToHost));
	printf("  %s\n", a == b ? "output identical to rsx_scatter2_kernel's (the whole array, column 1)" : "OUTPUT DIFFERS from rsx_scatter2_kernel's");
	g_flags = keepf | SCATTER_DBG_NOSTORE;
	printf("  without global stores: %.3f ms\n", run8_once<C, true>(0, false, grid));
	g_flags = 0;
}

static u32 *d_mailbox;

template <typename C, bool TL, bool HANDOFF, int HEAD_AT = 8>
float run9_once(u32 shift, bool dump)
{
	const u32 ntiles = (u32)(n / C::TILE);
	if (!d_mailbox) {
		CK(hipMalloc(&d_mailbox, (size_t)C::RING * 256 * 64));
		CK(hipMemset(d_mailbox, 0, (size_t)C::RING * 256 * 64));
	}
	CK(hipMemsetAsync(d_status, 0, 256 + (size_t)ntiles * 256 * 4, 0));
	if (TL)
		CK(hipMemsetAsync(d_tl, 0, (size_t)ntiles * 16 * 8, 0));
	hipEvent_t e0, e1;
	CK(hipEventCreate(&e0));
	CK(hipEventCreate(&e1));
	KdfArgs<u32> ka{0, 0, 0};
	CK(hipEventRecord(e0, 0));
	hipLaunchKernelGGL((rsx_scatter9_kernel<u32, u32, C, TL, DIG_PLAIN, HANDOFF, HEAD_AT>), dim3(ntiles), dim3(C::BLOCK), 0, 0, d_in, d_out, ntiles, shift,
	                   d_hist + 256 * (shift / 8), (u32 *)((char *)d_status + 256), (u32 *)d_status, d_mailbox, ka, g_flags, d_tl);
	CK(hipGetLastError());
	CK(hipEventRecord(e1, 0));
	CK(hipEventSynchronize(e1));
	float ms;
	CK(hipEventElapsedTime(&ms, e0, e1));
	CK(hipEventDestroy(e0));
	CK(hipEventDestroy(e1));
	if (TL && dump) {
		std::vector<u64> tl(ntiles * 16);
		CK(hipMemcpy(tl.data(), d_tl, (size_t)ntiles * 16 * 8, hipMemcpyDeviceToHost));
		double a = 0, lay = 0, ch = 0, st = 0, wo = 0, depth = 0, life = 0, w1 = 0, w2 = 0, w3 = 0;
		for (u64 t = 0; t < ntiles; ++t) {
			const u64 *r = &tl[t * 16];
			a += (double)(r[1] - r[0]);
			lay += (double)(r[2] - r[1]);
			ch += (double)(r[3] - r[2]);
			st += (double)(r[4] - r[2]);
			wo += (double)(r[5] - r[4]);
			life += (double)(r[5] - r[0]);
			depth += r[12];
			w1 += (double)(r[6] - r[4]);
			w2 += (double)(r[7] - r[6]) + (double)(r[5] - r[8]);
			w3 += (double)(r[8] - r[7]);
		}
		printf("  per tile: load + count %7.0f | layout %6.0f | stage %7.0f (chain %6.0f, depth %.1f, inside) | write-out %6.0f (records %5.0f, runs %5.0f, first atoms %5.0f; wave 0) | lifetime %7.0f\n",
		       a / ntiles, lay / ntiles, st / ntiles, ch / ntiles, depth / ntiles, wo / ntiles, w1 / ntiles, w2 / ntiles, w3 / ntiles, life / ntiles);
	}
	return ms;
}

template <typename C, bool HANDOFF, int HEAD_AT = 8>
void bench9(const char *name)
{
	run9_once<C, false, HANDOFF, HEAD_AT>(0, false);
	float best = 1e9, sum = 0;
	const int reps = 9;
	for (int i = 0; i < reps; ++i) {
		float ms = run9_once<C, false, HANDOFF, HEAD_AT>(8 * (i % 4), false);
		best = std::min(best, ms);
		sum += ms;
	}
	printf("%-22s tile %6d, lds %6zu B: avg %.3f ms best %.3f ms  -> %.0f GB/s (algorithmic 8 B/key)\n", name, C::TILE,
	       sizeof(Sc9Smem<u32, u32, C, HANDOFF>), sum / reps, best, n * 8.0 / (best * 1e-3) / 1e9);
	run9_once<C, true, HANDOFF, HEAD_AT>(0, true);
	std::vector<u32> a(n), b(n);
	for (u32 sh = 0; sh < 32; sh += 8) {
		run9_once<C, false, HANDOFF, HEAD_AT>(sh, false);
		CK(hipMemcpy(a.data(), d_out, n * 4, hipMemcpyDeviceToHost));
		run2_once<Sc2Cfg<u32, NoVal>, false, false>(sh, false, 1);
		CK(hipMemcpy(b.data(), d_out, n * 4, hipMemcpyDeviceToHost));
		printf("  column %u: %s\n", sh / 8, a == b ? "output identical to rsx_scatter2_kernel's (the whole array)" : "OUTPUT DIFFERS from rsx_scatter2_kernel's");
	}
	std::vector<u32> mb((size_t)C::RING * 256 * 16);
	CK(hipMemcpy(mb.data(), d_mailbox, mb.size() * 4, hipMemcpyDeviceToHost));
	size_t left = 0;
	for (size_t i = 15; i < mb.size(); i += 16)
		left += mb[i] != 0;
	printf("  records left unread: %zu\n", left);
}

template <typename C, bool TL, bool CHAIN_FIRST, bool NEXT_HIST = false>
float run10_once(u32 shift, bool dump)
{
	const u32 ntiles = (u32)(n / C::TILE);
	CK(hipMemsetAsync(d_status, 0, 256 + (size_t)ntiles * 256 * 4, 0));
	if (TL)
		CK(hipMemsetAsync(d_tl, 0, (size_t)ntiles * 16 * 8, 0));
	hipEvent_t e0, e1;
	CK(hipEventCreate(&e0));
	CK(hipEventCreate(&e1));
	KdfArgs<u32> ka{0, 0, 0};
	CK(hipEventRecord(e0, 0));
	hipLaunchKernelGGL((rsx_scatter10_kernel<u32, u32, C, TL, DIG_PLAIN, CHAIN_FIRST, NEXT_HIST>), dim3(ntiles), dim3(C::BLOCK), 0, 0, d_in, d_out, ntiles,
	                   shift, d_hist + 256 * (shift / 8), (u32 *)((char *)d_status + 256), (u32 *)d_status, ka, g_flags, d_tl,
	                   (unsigned long long *)(d_hist + 256 * 4));
	CK(hipGetLastError());
	CK(hipEventRecord(e1, 0));
	CK(hipEventSynchronize(e1));
	float ms;
	CK(hipEventElapsedTime(&ms, e0, e1));
	CK(hipEventDestroy(e0));
	CK(hipEventDestroy(e1));
	if (TL && dump) {
		std::vector<u64> tl(ntiles * 16);
		CK(hipMemcpy(tl.data(), d_tl, (size_t)ntiles * 16 * 8, hipMemcpyDeviceToHost));
		double a = 0, lay = 0, ch = 0, st = 0, wo = 0, depth = 0, life = 0, own = 0;
		for (u64 t = 0; t < ntiles; ++t) {
			const u64 *r = &tl[t * 16];
			a += (double)(r[1] - r[0]);
			lay += (double)(r[2] - r[1]);
			ch += (double)(r[3] - r[2]);
			st += (double)(r[4] - r[2]);
			wo += (double)(r[5] - r[4]);
			life += (double)(r[5] - r[0]);
			depth += r[12];
			own += r[6] ? (double)(r[6] - r[2]) : 0;
		}
		printf("  per tile: load + rank %7.0f | layout %6.0f | stage %7.0f (chain done at %6.0f, depth %.1f; wave 0 staged at %6.0f) | write-out %6.0f | lifetime %7.0f\n",
		       a / ntiles, lay / ntiles, st / ntiles, ch / ntiles, depth / ntiles, own / ntiles, wo / ntiles, life / ntiles);
	}
	return ms;
}

template <typename C, bool CHAIN_FIRST, bool NEXT_HIST = false>
void bench10(const char *name)
{
	run10_once<C, false, CHAIN_FIRST, NEXT_HIST>(0, false);
	float best = 1e9, sum = 0;
	const int reps = 9;
	for (int i = 0; i < reps; ++i) {
		float ms = run10_once<C, false, CHAIN_FIRST, NEXT_HIST>(8 * (i % 4), false);
		best = std::min(best, ms);
		sum += ms;
	}
	printf("%-30s tile %6d, lds %6zu B: avg %.3f ms best %.3f ms  -> %.0f GB/s (algorithmic 8 B/key)\n", name, C::TILE,
	       sizeof(Sc10Smem<u32, u32, C>), sum / reps, best, n * 8.0 / (best * 1e-3) / 1e9);
	run10_once<C, true, CHAIN_FIRST, NEXT_HIST>(0, true);
	std::vector<u32> a(n), b(n);
	run10_once<C, false, CHAIN_FIRST, NEXT_HIST>(8, false);
	CK(hipMemcpy(a.data(), d_out, n * 4, hipMemcpyDeviceToHost));
	run2_once<Sc2Cfg<u32, NoVal>, false, false>(8, false, 1);
	CK(hipMemcpy(b.data(), d_out, n * 4, hipMemcpyDeviceToHost));
	printf("  %s\n", a == b ? "output identical to rsx_scatter2_kernel's (the whole array, column 1)" : "OUTPUT DIFFERS from rsx_scatter2_kernel's");
	g_flags = SCATTER_DBG_NOSTORE;
	printf("  without global stores: %.3f ms\n", run10_once<C, true, CHAIN_FIRST, NEXT_HIST>(0, false));
	g_flags = 0;
}

template <typename C, bool HOTV = false>
void bench2(const char *name, u32 tps)
{
	run2_once<C, false, HOTV>(0, false, tps);
	float best = 1e9, sum = 0;
	const int reps = 5;
	for (int i = 0; i < reps; ++i) {
		float ms = run2_once<C, false, HOTV>(8 * (i % 4), false, tps);
		best = std::min(best, ms);
		sum += ms;
	}
	printf("%-12s tps %u tile %6d lds %6zu B: avg %.3f ms best %.3f ms  -> %.0f GB/s (algorithmic 8 B/key)\n", name, tps, C::TILE,
	       sizeof(Sc2Smem<u32, NoVal, u32, C>), sum / reps, best, n * 8.0 / (best * 1e-3) / 1e9);
	run2_once<C, true, HOTV>(0, true, tps);
}

template <typename SH>
void bench(const char *name, u32 tps)
{
	typedef ScatterCfg<u32, NoVal, SH> C;
	g_tps = tps;
	run_once<SH, false>(0, false);
	float best = 1e9, sum = 0;
	const int reps = 5;
	for (int i = 0; i < reps; ++i) {
		float ms = run_once<SH, false>(8 * (i % 4), false);
		best = std::min(best, ms);
		sum += ms;
	}
	printf("%-10s tps %u tile %6d lds %6zu B: avg %.3f ms best %.3f ms  -> %.0f GB/s (algorithmic 8 B/key)\n", name, tps, C::TILE,
	       sizeof(ScatterSmem<u32, NoVal, u32, SH>), sum / reps, best, n * 8.0 / (best * 1e-3) / 1e9);
	run_once<SH, true>(0, true);
}

// every aligned block of 256 keys holds each digit once in every byte column: a tile's runs are all 128 keys and start
// on 512-byte boundaries (an upper bound for what aligned runs could give)
__device__ inline u32 mix15(u32 x, u32 c, u32 m1, u32 m2)
{
	x = (x ^ c) & 0x7FFFu;   // every step is a bijection on 15 bits
	x ^= x >> 7;
	x = (x * m1) & 0x7FFFu;
	x ^= x >> 5;
	x = (x * m2) & 0x7FFFu;
	x ^= x >> 9;
	return x;
}

// mode 1: every tile of 32 Ki keys holds each digit exactly 128 times in every byte column, in scrambled order: all runs are
// 512 bytes and start on 512-byte boundaries.  mode 2: the same with run lengths 128 +- 16 j (64-byte aligned starts only);
// modes 3 and 4: 128 +- 4 j and 128 +- 8 j (16- and 32-byte aligned starts).
__global__ void sawtooth_kernel(u32 *a, u64 n)
{
	for (u64 i = blockIdx.x * (u64)blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x)
		a[i] = (u32)(i % 1000003ull * 4099ull % (1ull << 31));
}

__global__ void balanced_digits_kernel(u32 *a, u64 n, u32 mode)
{
	for (u64 i = blockIdx.x * (u64)blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) {
		const u32 t = (u32)(i >> 15) * 2654435761u, w = (u32)i & 0x7FFFu;
		u32 k = 0;
		const u32 m1[4] = {0x2545u, 0x1b0du, 0x6a6bu, 0x4f35u}, m2[4] = {0x5bd1u, 0x3c6fu, 0x0e99u, 0x7ab3u};
		for (int b = 0; b < 4; ++b) {
			const u32 x = mix15(w, t >> (4 * b + 1), m1[b], m2[b]);
			u32 d = x >> 7;
			if (mode == 2) {
				// slots of 16 keys (2048 per tile) dealt to digits unevenly: digit pairs (2m, 2m+1) get 8 + s and 8 - s slots
				const u32 slot = x >> 4, pair = slot >> 4, in = slot & 15u, s = (pair * 5u + (t >> 9)) % 7u;
				d = 2 * pair + (in < 8 + s ? 0 : 1);
			}
			if (mode == 3) {
				// slots of 4 keys: run lengths 128 +- 4 j, starts aligned to 16 bytes only
				const u32 slot = x >> 2, pair = slot >> 6, in = slot & 63u, s = (pair * 5u + (t >> 9)) % 7u;
				d = 2 * pair + (in < 32 + s ? 0 : 1);
			}
			if (mode == 4) {
				// slots of 8 keys: starts aligned to 32 bytes
				const u32 slot = x >> 3, pair = slot >> 5, in = slot & 31u, s = (pair * 5u + (t >> 9)) % 7u;
				d = 2 * pair + (in < 16 + s ? 0 : 1);
			}
			k |= d << (8 * b);
		}
		a[i] = k;
	}
}

int main(int argc, char **argv)
{
	const int log2n = argc > 1 ? atoi(argv[1]) : 28;
	n = (size_t)1 << log2n;
	CK(hipMalloc(&d_in, n * 4));
	CK(hipMalloc(&d_out, n * 4));
	CK(hipMalloc(&d_hist, 8 * 256 * 8));
	CK(hipMalloc(&d_flag, 256));
	CK(hipMalloc(&d_plan, sizeof(Plan)));
	CK(hipMalloc(&d_status, 256 + (n / 1024 + 1) * 256 * 4));
	CK(hipMalloc(&d_tl, (n / 1024 + 1) * 16 * 8));
	hipLaunchKernelGGL((rsx_fill_splitmix_kernel<u32>), dim3(2048), dim3(256), 0, 0, d_in, (u64)n, 1ull, ~0ull, 0ull);
	if (argc > 2 && atoi(argv[2]) == 5) {
		printf("four values per byte (keys & 0x03030303)\n");
		hipLaunchKernelGGL((rsx_fill_splitmix_kernel<u32>), dim3(2048), dim3(256), 0, 0, d_in, (u64)n, 1ull, 0x03030303ull, 0ull);
	} else if (argc > 2 && atoi(argv[2]) == 6) {
		printf("sawtooth (i %% 1000003 * 4099 %% 2^31: digits constant locally, flat globally)\n");
		hipLaunchKernelGGL(sawtooth_kernel, dim3(2048), dim3(256), 0, 0, d_in, (u64)n);
	} else if (argc > 2) {
		printf("balanced digits, mode %d\n", atoi(argv[2]));
		hipLaunchKernelGGL(balanced_digits_kernel, dim3(2048), dim3(256), 0, 0, d_in, (u64)n, (u32)atoi(argv[2]));
	}
	CK(hipMemset(d_hist, 0, 8 * 256 * 8));
	CK(hipMemset(d_flag, 0, 256));
	KdfArgs<u32> ka{0, 0, 0};
	u32 *d_part;
	CK(hipMalloc(&d_part, 512 * 1024 * 4));
	hipLaunchKernelGGL((rsx_hist_kernel<u32>), dim3(512), dim3(HistCfg<u32>::BLOCK), 0, 0, d_in, (u64)n, d_part, d_flag, ka, ~0u, (u64 *)nullptr);
	hipLaunchKernelGGL(rsx_hist_reduce_kernel, dim3(4, HIST_REDUCE_SPLIT), dim3(256), 0, 0, (const u32 *)d_part, d_hist, 512u, 1024u);
	hipLaunchKernelGGL((rsx_plan_kernel<u32>), dim3(4), dim3(256), 0, 0, d_in, (u64)n, d_hist, ka, d_flag + 8, d_flag + 32);
	CK(hipDeviceSynchronize());
	printf("n = 2^%d u32 keys\n", log2n);
	bench2<Sc2Cfg<u32, NoVal>>("v2 default", 1);
	if (getenv("RSX_PROBE_V11")) {
		bench11<Sc11Cfg<u32, 8>>("v11 pipelined, cursors", 256);
		g_flags = SCATTER_DBG_LINEAR;
		bench2<Sc2Cfg<u32, NoVal>>("v2, staged tile written back to its own place", 1);
		g_flags = 0;
		bench8<Sc8Cfg<u32, 8>>("v8 pipelined LB 8", 256);
		bench2<Sc2Cfg<u32, NoVal>>("v2 default", 1);
		return 0;
	}
	if (getenv("RSX_PROBE_LB")) {
		bench2<Sc2Cfg<u32, NoVal, 16, 1, 4>>("v2 LB 4", 1);
		bench2<Sc2Cfg<u32, NoVal, 16, 1, 12>>("v2 LB 12", 1);
		bench2<Sc2Cfg<u32, NoVal, 16, 1, 16>>("v2 LB 16", 1);
		bench2<Sc2Cfg<u32, NoVal>>("v2 default", 1);
	}
	if (argc > 2 && atoi(argv[2]) == 5) {
		u32 hd[9];
		CK(hipMemcpy(hd, d_flag + 32, sizeof hd, hipMemcpyDeviceToHost));
		printf("hotd: %08x %08x %08x %08x valid %08x\n", hd[0], hd[1], hd[2], hd[3], hd[8]);
		bench2<Sc2Cfg<u32, NoVal>, true>("v2 HOT", 1);
		g_flags = SCATTER_DBG_NOSTORE;
		printf("-- v2 HOT, no global stores: %.3f ms\n", run2_once<Sc2Cfg<u32, NoVal>, true, true>(0, true, 1));
		printf("-- v2 plain, no global stores: %.3f ms\n", run2_once<Sc2Cfg<u32, NoVal>, true, false>(0, true, 1));
		g_flags = 0;
	}
	g_flags = SCATTER_ELEM_LOADS;
	bench2<Sc2Cfg<u32, NoVal>>("v2 elem loads", 1);
	g_flags = 0;
	if (getenv("RSX_PROBE_NARROW")) {
		// what a pass costs that writes only the two low bytes of every key (a level-2 pass whose leaves need nothing else)
		typedef Sc2Cfg<u32, NoVal> C;
		const u64 tiles = (n + C::TILE - 1) / C::TILE;
		for (int rep = 0; rep < 6; ++rep) {
			CK(hipMemsetAsync(d_status, 0, 256 + tiles * 256 * 4, 0));
			hipEvent_t e0, e1;
			CK(hipEventCreate(&e0));
			CK(hipEventCreate(&e1));
			CK(hipEventRecord(e0, 0));
			if (rep & 1)
				hipLaunchKernelGGL((rsx_scatter2_kernel<u32, NoVal, u32, C, false, DIG_PLAIN, false, uint16_t>), dim3((unsigned)tiles), dim3(C::BLOCK),
				                   0, 0, d_in, (uint16_t *)d_out, (const NoVal *)nullptr, (NoVal *)nullptr, (u64)n, 16u, d_hist + 256 * 2, 1u,
				                   (u32 *)((char *)d_status + 256), (u32 *)d_status, ka, (u32)SCATTER_ELEM_LOADS, (u64 *)nullptr,
				                   (const Plan *)nullptr, 0u, 0u, (const u32 *)nullptr);
			else
				hipLaunchKernelGGL((rsx_scatter2_kernel<u32, NoVal, u32, C, false, DIG_PLAIN, false>), dim3((unsigned)tiles), dim3(C::BLOCK),
				                   0, 0, d_in, d_out, (const NoVal *)nullptr, (NoVal *)nullptr, (u64)n, 16u, d_hist + 256 * 2, 1u,
				                   (u32 *)((char *)d_status + 256), (u32 *)d_status, ka, (u32)SCATTER_ELEM_LOADS, (u64 *)nullptr,
				                   (const Plan *)nullptr, 0u, 0u, (const u32 *)nullptr);
			CK(hipGetLastError());
			CK(hipEventRecord(e1, 0));
			CK(hipEventSynchronize(e1));
			float ms;
			CK(hipEventElapsedTime(&ms, e0, e1));
			printf("pass by column 2, keys written as %s: %.3f ms\n", (rep & 1) ? "their low 16 bits (u16)" : "u32", ms);
		}
		return 0;
	}
	if (getenv("RSX_PROBE_ALL"))
		bench7<Sc7Cfg<u32>>("v7 2 WG/CU, re-ranked windows");
	if (getenv("RSX_PROBE_ONE_ATOMIC") || getenv("RSX_PROBE_ALL")) {
		bench9<Sc9Cfg<u32>, false>("v9 plain");
		bench10<Sc10Cfg<u32>, true>("v10 one atomic, chain first");
		bench10<Sc10Cfg<u32>, false>("v10 one atomic, stage first");
		bench10<Sc10Cfg<u32>, true, true>("v10 + next column's histogram");
	{
		// the fused histogram of column 1 (shift 0 run) against the one the histogram kernel made
		CK(hipMemset(d_hist + 256 * 4, 0, 256 * 8));
		run10_once<Sc10Cfg<u32>, false, true, true>(0, false);
		std::vector<u64> a(256), b(256);
		CK(hipMemcpy(a.data(), d_hist + 256 * 4, 256 * 8, hipMemcpyDeviceToHost));
		CK(hipMemcpy(b.data(), d_hist + 256 * 1, 256 * 8, hipMemcpyDeviceToHost));
		// (d_hist holds exclusive offsets after the plan kernel: compare differences)
		bool same = true;
		for (int i = 0; i + 1 < 256; ++i)
			same &= a[i] == b[i + 1] - b[i];
		printf("  fused histogram of column 1 %s the histogram kernel's\n", same ? "equals" : "DIFFERS from");
	}
	bench10<Sc10Cfg<u32, 8, 8>, true>("v10 8 waves, 2 WG/CU, chain first");
		bench10<Sc10Cfg<u32, 8, 8>, false>("v10 8 waves, 2 WG/CU, stage first");
		bench10<Sc10Cfg<u32, 16, 8>, true>("v10 8 waves, 2 WG/CU, LB 16");
		bench9<Sc9Cfg<u32>, false>("v9 plain");
	}
	if (getenv("RSX_PROBE_HANDOFF") || getenv("RSX_PROBE_ALL")) {
		bench9<Sc9Cfg<u32>, false>("v9 plain");
		bench9<Sc9Cfg<u32>, true, 8>("v9 handed on, atoms last");
		{
			const struct { const char *name; u32 f; } dbg[] = {
				{"no deposit, no read", SC9_DBG_NODEPOSIT | SC9_DBG_NOSPIN | SC9_DBG_NOREAD},
				{"no deposit, read unchecked", SC9_DBG_NODEPOSIT | SC9_DBG_NOSPIN},
				{"plain deposit, no read", SC9_DBG_PLAINDEPOSIT | SC9_DBG_NOSPIN | SC9_DBG_NOREAD},
				{"sc1 deposit, no read", SC9_DBG_NOSPIN | SC9_DBG_NOREAD},
			};
			for (const auto &v : dbg) {
				g_flags = v.f;
				float best = 1e9;
				for (int i = 0; i < 6; ++i)
					best = std::min(best, run9_once<Sc9Cfg<u32>, false, true, 8>(8 * (i % 4), false));
				printf("v9 timing only (wrong output), %s: best %.3f ms\n", v.name, best);
				run9_once<Sc9Cfg<u32>, true, true, 8>(0, true);
				g_flags = 0;
				CK(hipMemset(d_mailbox, 0, (size_t)Sc9Cfg<u32>::RING * 256 * 64));
			}
		}
	}
	if (getenv("RSX_PROBE_ALL")) {
		bench8<Sc8Cfg<u32, 8>>("v8 pipelined LB 8", 256);
		bench8<Sc8Cfg<u32, 16>>("v8 pipelined LB 16", 256);
		bench8<Sc8Cfg<u32, 24>>("v8 pipelined LB 24", 256);
	}
	if (getenv("RSX_PROBE_READ_AHEAD"))
		bench_read_ahead();
	for (u32 lr = 0; lr <= 5 && getenv("RSX_PROBE_XCD_RUNS"); ++lr) {
		// tiles by workgroup index: runs of 2^lr consecutive tiles on one XCD (their ragged run ends meet in one L2)
		g_flags = SCATTER_DBG_XCD_RUNS | (lr << SCATTER_XCD_RUN_SHIFT);
		char nm[64];
		snprintf(nm, sizeof nm, "v2 xcd runs %u", 1u << lr);
		bench2<Sc2Cfg<u32, NoVal>>(nm, 1);
		std::vector<u32> a(n), b(n);
		run2_once<Sc2Cfg<u32, NoVal>, false, false>(8, false, 1);
		CK(hipMemcpy(a.data(), d_out, n * 4, hipMemcpyDeviceToHost));
		g_flags = 0;
		run2_once<Sc2Cfg<u32, NoVal>, false, false>(8, false, 1);
		CK(hipMemcpy(b.data(), d_out, n * 4, hipMemcpyDeviceToHost));
		printf("  %s\n", a == b ? "output identical" : "OUTPUT DIFFERS");
	}
	if (getenv("RSX_PROBE_ALL")) {
		// round 2's structural experiments (each a header of its own next to this file, with its numbers)
		bench3<Sc3Cfg<u32, 8, 8, 48>>("v3 2 WG/CU, 2 windows");
		bench4<Sc4Cfg<u32>>("v4 pairs");
		bench5<Sc5Cfg<u32>>("v5 persistent + prefetch", 256);
		g_flags = SCATTER_DBG_LINEAR;
		bench5<Sc5Cfg<u32>>("v5 persistent, no prefetch", 256);
		g_flags = 0;
		bench6<Sc6Cfg<u32>>("v6 digit waves");
	}
	// what the pass costs without its global stores (the keys are read, counted, chained, staged and read back)
	g_flags = SCATTER_DBG_NOSTORE;
	printf("-- v2 default, no global stores: %.3f ms\n", run2_once<Sc2Cfg<u32, NoVal>, true, false>(0, true, 1));
	g_flags = 0;
	if (argc > 2)
		return 0;
	bench<DefaultShape<u32, NoVal, RANK_TABLE>>("v1 table", 4);
	// correctness of v2: digits of the output must be non-decreasing and the multiset preserved (checked via sum)
	{
		typedef Sc2Cfg<u32, NoVal> C;
		run2_once<C, false>(0, false, 1);
		std::vector<u32> out(1 << 20), in(1 << 20);
		CK(hipMemcpy(out.data(), d_out, out.size() * 4, hipMemcpyDeviceToHost));
		size_t bad = 0;
		for (size_t i = 1; i < out.size(); ++i)
			bad += (out[i - 1] & 0xFF) > (out[i] & 0xFF);
		printf("v2 check: digit order violations in the first 2^20 outputs: %zu\n", bad);
	}
	return 0;
}
