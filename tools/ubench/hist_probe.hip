// hist_probe: the histogram kernel (radix_sorting_amd/csrc/rsx_hist.hpp) against round 1's (rsx_hist_r1.hpp) and against a
// bare read of the same bytes with the same launch shape; launch shapes (workgroup size, loads in flight, grid).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I radix_sorting_amd/csrc tools/ubench/hist_probe.hip -o tools/ubench/hist_probe.bin
// Run:   hist_probe.bin [log2 n = 28] [key bytes = 4 | 8] [mask, hex: the keys are splitmix64 & mask]
#include "rsx_kernels.hpp"
#include "rsx_hist_r1.hpp"

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

static void *d_in;
static u64 *d_hist;
static u32 *d_part;
static u32 *d_flag;
static size_t n;
static u32 g_colmask = ~0u;
static std::vector<u64> g_ref;   // the first variant's histogram: every other variant must reproduce it

// a read of the same bytes with the kernel's launch shape and loop, nothing counted (the floor)
template <typename KT, int BLOCK, int U> __global__ __launch_bounds__(BLOCK) void read_only_kernel(const KT *src, u64 n, u32 *out)
{
	constexpr int VEC = 16 / sizeof(KT);
	typedef KT vec_t __attribute__((ext_vector_type(VEC)));
	const vec_t *vsrc = (const vec_t *)src;
	const u64 nvec = n / VEC, stride = (u64)gridDim.x * (BLOCK * U);
	KT x = 0;
	for (u64 vb = (u64)blockIdx.x * (BLOCK * U); vb < nvec; vb += stride) {
		vec_t raw[U];
#pragma unroll
		for (int u = 0; u < U; ++u)
			if (vb + threadIdx.x + (u64)u * BLOCK < nvec)
				raw[u] = vsrc[vb + threadIdx.x + (u64)u * BLOCK];
#pragma unroll
		for (int u = 0; u < U; ++u)
			if (vb + threadIdx.x + (u64)u * BLOCK < nvec)
#pragma unroll
				for (int e = 0; e < VEC; ++e)
					x ^= raw[u][e];
	}
	if (x == (KT)0x12345)
		out[threadIdx.x] = 1;
}

template <typename KT> void check_and_print(const char *name, unsigned grid, int block, int u, float best)
{
	constexpr int WC = sizeof(KT);
	std::vector<u64> h(WC * 256);
	CK(hipMemcpy(h.data(), d_hist, h.size() * 8, hipMemcpyDeviceToHost));
	u64 tot = 0;
	for (int i = 0; i < 256; ++i)
		tot += h[i];
	const char *verdict = "";
	if (g_colmask == ~0u) {
		if (g_ref.empty())
			g_ref = h, verdict = "(reference)";
		else
			verdict = h == g_ref ? "same counts" : "COUNTS DIFFER";
	}
	printf("%-30s grid %5u block %4d U %d: %.3f ms  %5.0f GB/s  column 0 total %llu  %s\n", name, grid, block, u, best,
	       n * (double)sizeof(KT) / (best * 1e-3) / 1e9, (unsigned long long)tot, verdict);
}

template <typename KT, typename F> float time_it(F launch)
{
	float best = 1e9;
	// RSX_PROBE_COLD=1: a gigabyte of other memory is written between the timed launches -- nothing of the input is left in the
	// 256 MB Infinity Cache, as inside a sort, whose histogram kernel reads keys nobody has read before
	static char *thrash = nullptr;
	static const bool cold = getenv("RSX_PROBE_COLD") != nullptr;
	if (cold && !thrash)
		CK(hipMalloc(&thrash, (size_t)1 << 30));
	for (int i = 0; i < 6; ++i) {
		if (cold)
			CK(hipMemsetAsync(thrash, i, (size_t)1 << 30, 0));
		CK(hipMemsetAsync(d_hist, 0, sizeof(KT) * 256 * 8, 0));
		hipEvent_t e0, e1;
		CK(hipEventCreate(&e0));
		CK(hipEventCreate(&e1));
		CK(hipEventRecord(e0, 0));
		launch();
		CK(hipGetLastError());
		CK(hipEventRecord(e1, 0));
		CK(hipEventSynchronize(e1));
		float ms;
		CK(hipEventElapsedTime(&ms, e0, e1));
		best = std::min(best, ms);
		CK(hipEventDestroy(e0));
		CK(hipEventDestroy(e1));
	}
	return best;
}

template <typename KT, typename C> void bench_r1(const char *name, unsigned grid)
{
	KdfArgs<KT> ka{0, 0, 0};
	const float best = time_it<KT>([&] {
		hipLaunchKernelGGL((rsx_hist_r1_kernel<KT, C>), dim3(grid), dim3(C::BLOCK), 0, 0, (const KT *)d_in, (u64)n, d_part, d_flag, ka, 1u,
		                   grid, (u64)n, g_colmask);
		hipLaunchKernelGGL(rsx_hist_r1_reduce_kernel, dim3(sizeof(KT), HIST_R1_REDUCE_SPLIT), dim3(256), 0, 0, (const u32 *)d_part, d_hist,
		                   grid, (u32)sizeof(KT) * 256u);
	});
	check_and_print<KT>(name, grid, C::BLOCK, C::U, best);
}

template <typename KT, typename C, int MODE = HIST_PLAIN> void bench_new(const char *name, unsigned grid)
{
	KdfArgs<KT> ka{0, 0, 0};
	const float best = time_it<KT>([&] {
		hipLaunchKernelGGL((rsx_hist_kernel<KT, C, MODE>), dim3(grid), dim3(C::BLOCK), 0, 0, (const KT *)d_in, (u64)n, d_part, d_flag, ka,
		                   g_colmask, (u64 *)nullptr);
		hipLaunchKernelGGL(rsx_hist_reduce_kernel, dim3(sizeof(KT), HIST_REDUCE_SPLIT), dim3(256), 0, 0, (const u32 *)d_part, d_hist, grid,
		                   (u32)sizeof(KT) * 256u);
	});
	check_and_print<KT>(name, grid, C::BLOCK, C::U, best);
}

template <typename KT, int BLOCK, int U> void bench_read(unsigned grid)
{
	const float best = time_it<KT>([&] {
		hipLaunchKernelGGL((read_only_kernel<KT, BLOCK, U>), dim3(grid), dim3(BLOCK), 0, 0, (const KT *)d_in, (u64)n, d_part);
	});
	printf("%-30s grid %5u block %4d U %d: %.3f ms  %5.0f GB/s\n", "read only", grid, BLOCK, U, best,
	       n * (double)sizeof(KT) / (best * 1e-3) / 1e9);
}

template <typename KT> void run(u64 mask)
{
	CK(hipMalloc(&d_in, n * sizeof(KT)));
	hipLaunchKernelGGL((rsx_fill_splitmix_kernel<KT>), dim3(2048), dim3(256), 0, 0, (KT *)d_in, (u64)n, 1ull, mask, 0ull);
	CK(hipDeviceSynchronize());
	printf("n = %zu keys of %zu bytes, mask %016llx\n", n, sizeof(KT), (unsigned long long)mask);
	for (int rep = 0; rep < 2; ++rep) {
		bench_r1<KT, HistR1Cfg<KT, 1024, 2>>("round 1", 512);
		bench_new<KT, HistCfg<KT, 1024, 2>>("new, plain", 512);
		bench_new<KT, HistCfg<KT, 1024, 2>, HIST_GENERIC>("new, generic KDF", 512);
		bench_new<KT, HistCfg<KT, 1024, 1>>("new, plain", 512);
		bench_new<KT, HistCfg<KT, 1024, 3>>("new, plain", 512);
		bench_new<KT, HistCfg<KT, 1024, 4>>("new, plain", 512);
		bench_new<KT, HistCfg<KT, 512, 2>>("new, plain", 1024);
		bench_new<KT, HistCfg<KT, 512, 4>>("new, plain", 1024);
		bench_new<KT, HistCfg<KT, 1024, 2>>("new, plain", 1024);
		bench_new<KT, HistCfg<KT, 1024, 2>>("new, plain", 2048);
		bench_read<KT, 1024, 2>(512);
		bench_read<KT, 1024, 4>(512);
		bench_read<KT, 1024, 4>(2048);
	}
	// how the time follows the number of LDS atomics per key
	for (u32 m : {0x7u, 0x3u, 0x1u}) {
		g_colmask = m;
		printf("columns counted: mask %x\n", m);
		bench_r1<KT, HistR1Cfg<KT, 1024, 2>>("round 1", 512);
		bench_new<KT, HistCfg<KT, 1024, 2>>("new", 512);
	}
	g_colmask = ~0u;
}

int main(int argc, char **argv)
{
	const int log2n = argc > 1 ? atoi(argv[1]) : 28;
	const int kb = argc > 2 ? atoi(argv[2]) : 4;
	const u64 mask = argc > 3 ? strtoull(argv[3], nullptr, 16) : ~0ull;
	n = (size_t)1 << log2n;
	CK(hipMalloc(&d_hist, 8 * 256 * 8));
	CK(hipMalloc(&d_flag, 64));
	CK(hipMalloc(&d_part, 4096 * 2048 * 4));
	CK(hipMemset(d_flag, 0, 64));
	if (kb == 8)
		run<u64>(mask);
	else
		run<u32>(mask);
	return 0;
}
