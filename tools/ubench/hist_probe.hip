// hist_probe: launch shapes of rsx_hist_kernel (workgroup size, loads in flight, grid).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I radix_sorting_amd/csrc tools/ubench/hist_probe.hip -o tools/ubench/hist_probe.bin
#include "rsx_kernels.hpp"

#include <algorithm>
#include <cstdio>
#include <cstdlib>

using namespace rsx;

#define CK(x)                                                                         \
	do {                                                                              \
		hipError_t e_ = (x);                                                          \
		if (e_ != hipSuccess) {                                                       \
			printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
			exit(1);                                                                  \
		}                                                                             \
	} while (0)

static u32 *d_in;
static u64 *d_hist;
static u32 *d_part;
static u32 *d_flag;
static size_t n;
static u32 g_colmask = ~0u;

template <typename C> void bench(const char *name, unsigned grid)
{
	KdfArgs<u32> ka{0, 0, 0};
	float best = 1e9;
	for (int i = 0; i < 6; ++i) {
		CK(hipMemsetAsync(d_hist, 0, 4 * 256 * 8, 0));
		hipEvent_t e0, e1;
		CK(hipEventCreate(&e0));
		CK(hipEventCreate(&e1));
		CK(hipEventRecord(e0, 0));
		hipLaunchKernelGGL((rsx_hist_kernel<u32, C>), dim3(grid), dim3(C::BLOCK), 0, 0, d_in, (u64)n, d_part, d_flag, ka, 1u, grid, (u64)n, g_colmask);
		hipLaunchKernelGGL(rsx_hist_reduce_kernel, dim3(4, HIST_REDUCE_SPLIT), dim3(256), 0, 0, (const u32 *)d_part, d_hist, grid, 1024u);
		CK(hipGetLastError());
		CK(hipEventRecord(e1, 0));
		CK(hipEventSynchronize(e1));
		float ms;
		CK(hipEventElapsedTime(&ms, e0, e1));
		best = std::min(best, ms);
	}
	u64 h[1024];
	CK(hipMemcpy(h, d_hist, sizeof(h), hipMemcpyDeviceToHost));
	u64 tot = 0, chk = 0;
	for (int i = 0; i < 256; ++i)
		tot += h[i];
	for (int i = 0; i < 1024; ++i)
		chk = chk * 1315423911ull + h[i];
	printf("%-28s grid %5u block %4d U %d R %d: %.3f ms  %.0f GB/s  (column 0 total %llu, checksum %016llx)\n", name, grid, C::BLOCK, C::U,
	       C::R, best, n * 4.0 / (best * 1e-3) / 1e9, (unsigned long long)tot, (unsigned long long)chk);
}

int main(int argc, char **argv)
{
	const int log2n = argc > 1 ? atoi(argv[1]) : 28;
	n = (size_t)1 << log2n;
	CK(hipMalloc(&d_in, n * 4));
	CK(hipMalloc(&d_hist, 8 * 256 * 8));
	CK(hipMalloc(&d_flag, 64));
	CK(hipMalloc(&d_part, 4096 * 1024 * 4));
	hipLaunchKernelGGL((rsx_fill_splitmix_kernel<u32>), dim3(2048), dim3(256), 0, 0, d_in, (u64)n, 1ull, ~0ull, 0ull);
	CK(hipMemset(d_flag, 0, 64));
	CK(hipDeviceSynchronize());
	printf("n = 2^%d u32 keys\n", log2n);
	// how the time follows the number of LDS atomics per key, and lane-private stripes (R = 32: no bank conflicts, one
	// workgroup per CU)
	for (u32 m : {0xFu, 0x7u, 0x3u, 0x1u}) {
		g_colmask = m;
		printf("columns counted: mask %x\n", m);
		bench<HistCfg<u32, 1024, 2>>("block 1024 U2 R16", 512);
		bench<HistCfg<u32, 1024, 2, 32>>("block 1024 U2 R32", 256);
		bench<HistCfg<u32, 1024, 2, 8>>("block 1024 U2 R8", 512);
	}
	g_colmask = ~0u;
	if (argc > 2)
		return 0;
	for (int rep = 0; rep < 2; ++rep) {
		bench<HistCfg<u32, 1024, 1>>("block 1024 U1", 512);
		bench<HistCfg<u32, 1024, 2>>("block 1024 U2", 512);
		bench<HistCfg<u32, 1024, 3>>("block 1024 U3", 512);
		bench<HistCfg<u32, 1024, 4>>("block 1024 U4", 512);
		bench<HistCfg<u32, 1024, 8>>("block 1024 U8", 512);
		bench<HistCfg<u32, 1024, 2, 8>>("block 1024 U2 R8", 512);
		bench<HistCfg<u32, 1024, 2>>("block 1024 U2", 256);
		bench<HistCfg<u32, 1024, 2>>("block 1024 U2", 1024);
		bench<HistCfg<u32, 512, 2>>("block 512 U2", 1024);
		bench<HistCfg<u32, 256, 2, 8>>("block 256 U2 R8", 2048);
	}
	return 0;
}
