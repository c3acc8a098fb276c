// scatter_probe: tuning harness for rsx_scatter_kernel (tile shapes, phase timeline).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I radix_sorting_amd/csrc tools/ubench/scatter_probe.hip -o tools/ubench/scatter_probe.bin
#include "rsx_kernels.hpp"

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

template <typename SH, bool TL>
float run_once(u32 shift, bool timeline_dump)
{
	typedef ScatterCfg<u32, NoVal, SH> C;
	const u64 tiles = ((n + C::TILE - 1) / C::TILE + SH::SEGS - 1) / SH::SEGS * SH::SEGS;
	const size_t st_bytes = 256 + tiles * 256 * 4;
	CK(hipMemsetAsync(d_status, 0, st_bytes, 0));
	if (TL)
		CK(hipMemsetAsync(d_tl, 0, tiles * 16 * 8, 0));
	hipEvent_t e0, e1;
	CK(hipEventCreate(&e0));
	CK(hipEventCreate(&e1));
	KdfArgs<u32> ka{0, 0, 0};
	CK(hipEventRecord(e0, 0));
	hipLaunchKernelGGL((rsx_scatter_kernel<u32, NoVal, u32, SH, TL>), dim3((unsigned)tiles), dim3(C::BLOCK), 0, 0, d_in, d_out,
	                   (const NoVal *)nullptr, (NoVal *)nullptr, (u64)n, shift, d_hist + 256 * (shift / 8),
	                   (u32 *)((char *)d_status + 256), (u32 *)d_status, ka, 0u, (const uint8_t *)nullptr, d_tl);
	CK(hipGetLastError());
	CK(hipEventRecord(e1, 0));
	CK(hipEventSynchronize(e1));
	float ms;
	CK(hipEventElapsedTime(&ms, e0, e1));
	if (TL && timeline_dump) {
		std::vector<u64> tl(tiles * 16);
		CK(hipMemcpy(tl.data(), d_tl, tiles * 16 * 8, hipMemcpyDeviceToHost));
		const char *names[] = {"ticket+zero", "load wait", "rank", "sync1", "agg+lookback(d0)", "scan+sync2", "bases+sync3", "stage",
		                       "sync4", "writeout"};
		double sum[10] = {0};
		double depth = 0, maxdepth = 0;
		u64 tmin = ~0ull, tmax = 0;
		for (u64 t = 0; t < tiles; ++t) {
			const u64 *r = &tl[t * 16];
			for (int k = 0; k < 10; ++k)
				sum[k] += (double)(r[k + 1] - r[k]);
			depth += r[12];
			maxdepth = std::max(maxdepth, (double)r[12]);
			tmin = std::min(tmin, r[0]);
			tmax = std::max(tmax, r[10]);
		}
		double total = 0;
		for (int k = 0; k < 10; ++k)
			total += sum[k];
		printf("  timeline (avg s_memtime ticks per tile; kernel span %llu ticks, %.3f ms => %.1f MHz tick):\n",
		       (unsigned long long)(tmax - tmin), ms, (tmax - tmin) / (ms * 1e3));
		for (int k = 0; k < 10; ++k)
			printf("    %-20s %9.0f  (%4.1f%%)\n", names[k], sum[k] / tiles, 100.0 * sum[k] / total);
		printf("    total per tile       %9.0f ; look-back depth (digit 0): avg %.2f max %.0f\n", total / tiles, depth / tiles, maxdepth);
	}
	return ms;
}

template <typename SH>
void bench(const char *name)
{
	typedef ScatterCfg<u32, NoVal, SH> C;
	run_once<SH, false>(0, false);
	float best = 1e9, sum = 0;
	const int reps = 5;
	for (int i = 0; i < reps; ++i) {
		float ms = run_once<SH, false>(8 * (i % 4), false);
		best = std::min(best, ms);
		sum += ms;
	}
	printf("%-10s tile %6d lds %6zu B: avg %.3f ms best %.3f ms  -> %.0f GB/s (algorithmic 8 B/key)\n", name, C::TILE,
	       sizeof(ScatterSmem<u32, NoVal, u32, SH>), sum / reps, best, n * 8.0 / (best * 1e-3) / 1e9);
	run_once<SH, true>(0, true);
}

int main(int argc, char **argv)
{
	const int log2n = argc > 1 ? atoi(argv[1]) : 28;
	n = (size_t)1 << log2n;
	CK(hipMalloc(&d_in, n * 4));
	CK(hipMalloc(&d_out, n * 4));
	CK(hipMalloc(&d_hist, 8 * 256 * 8));
	CK(hipMalloc(&d_flag, 64));
	CK(hipMalloc(&d_plan, sizeof(Plan)));
	CK(hipMalloc(&d_status, 256 + (n / 1024 + 1) * 256 * 4));
	CK(hipMalloc(&d_tl, (n / 1024 + 1) * 16 * 8));
	hipLaunchKernelGGL((rsx_fill_splitmix_kernel<u32>), dim3(2048), dim3(256), 0, 0, d_in, (u64)n, 1ull, ~0ull, 0ull);
	CK(hipMemset(d_hist, 0, 8 * 256 * 8));
	CK(hipMemset(d_flag, 0, 64));
	KdfArgs<u32> ka{0, 0, 0};
	hipLaunchKernelGGL((rsx_hist_kernel<u32>), dim3(2048), dim3(256), 0, 0, d_in, (u64)n, d_hist, d_flag, ka);
	hipLaunchKernelGGL((rsx_plan_kernel<u32>), dim3(1), dim3(256), 0, 0, d_in, (u64)n, d_hist, d_flag, d_plan, ka);
	CK(hipDeviceSynchronize());
	printf("n = 2^%d u32 keys\n", log2n);
	bench<TileShape<8, 16, 8, 2, 1>>("8x16 s1");
	bench<TileShape<8, 16, 8, 2, 8>>("8x16 s8");
	bench<TileShape<8, 16, 8, 2, 32>>("8x16 s32");
	bench<TileShape<8, 16, 8, 2, 128>>("8x16 s128");
	bench<TileShape<8, 16, 4, 2, 128>>("8x16 s128 lb4");
	bench<TileShape<8, 16, 8, 2, 512>>("8x16 s512");
	return 0;
}
