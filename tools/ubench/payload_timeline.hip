// payload_timeline: the per-phase cycle table of the key + payload instantiations of rsx_scatter2_kernel (the `tl` stamps the
// keys-only kernel has in scatter_probe.hip): 2^28 pairs, uniform digits, one pass each of
//   <u32 keys, u32 payload>            (cfg 4 passes 1-2, pairs)
//   <u16 keys -> u8 keys, u32 payload> (cfg 4 pass 3, narrowed keys)
//   <u8 keys, u32 payload, no keys out>(cfg 4 pass 4)
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I radix_sorting_amd/csrc tools/ubench/payload_timeline.hip -o tools/ubench/payload_timeline.bin
#include "rsx_scatter2.hpp"

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

static size_t n;
static void *d_kin, *d_kout;
static u32 *d_vin, *d_vout;
static u64 *d_hist, *d_tl;
static void *d_status;

template <typename KT> __global__ void hist_col0(const KT *k, u64 n, u64 *h)
{
	__shared__ u32 s[256];
	s[threadIdx.x] = 0;
	__syncthreads();
	for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < n; i += (u64)gridDim.x * 256)
		atomicAdd(&s[(u32)k[i] & 0xFFu], 1u);
	__syncthreads();
	atomicAdd((unsigned long long *)&h[threadIdx.x], (unsigned long long)s[threadIdx.x]);
}
__global__ void scan256(u64 *h)
{
	if (threadIdx.x == 0) {
		u64 a = 0;
		for (int i = 0; i < 256; ++i) {
			const u64 c = h[i];
			h[i] = a;
			a += c;
		}
	}
}

template <typename KT, typename KTO>
void run(const char *name, u32 flags)
{
	typedef Sc2Cfg<KT, u32> C;
	const u64 tiles = (n + C::TILE - 1) / C::TILE;
	hipLaunchKernelGGL((rsx_fill_splitmix_kernel<KT>), dim3(2048), dim3(256), 0, 0, (KT *)d_kin, (u64)n, 7ull, ~0ull, 0ull);
	hipLaunchKernelGGL((rsx_iota_kernel<u32>), dim3(2048), dim3(256), 0, 0, d_vin, (u64)n);
	CK(hipMemset(d_hist, 0, 256 * 8));
	hipLaunchKernelGGL((hist_col0<KT>), dim3(1024), dim3(256), 0, 0, (const KT *)d_kin, (u64)n, d_hist);
	hipLaunchKernelGGL(scan256, dim3(1), dim3(64), 0, 0, d_hist);
	KdfArgs<KT> ka{0, 0, 0};
	float best = 1e9f;
	for (int rep = 0; rep < 4; ++rep) {
		CK(hipMemsetAsync(d_status, 0, 256 + tiles * 256 * 4, 0));
		CK(hipMemsetAsync(d_tl, 0, tiles * 16 * 8, 0));
		hipEvent_t e0, e1;
		CK(hipEventCreate(&e0));
		CK(hipEventCreate(&e1));
		CK(hipEventRecord(e0, 0));
		hipLaunchKernelGGL((rsx_scatter2_kernel<KT, u32, u32, C, true, DIG_GENERIC, false, KTO>), dim3((unsigned)tiles), dim3(C::BLOCK), 0, 0,
		                   (const KT *)d_kin, (KTO *)d_kout, (const u32 *)d_vin, d_vout, (u64)n, 0u, (const u64 *)d_hist, 1u,
		                   (u32 *)((char *)d_status + 256), (u32 *)d_status, ka, flags, d_tl);
		CK(hipGetLastError());
		CK(hipEventRecord(e1, 0));
		CK(hipEventSynchronize(e1));
		float ms;
		CK(hipEventElapsedTime(&ms, e0, e1));
		if (ms < best)
			best = ms;
	}
	std::vector<u64> tl(tiles * 16);
	CK(hipMemcpy(tl.data(), d_tl, tiles * 16 * 8, hipMemcpyDeviceToHost));
	double a = 0, lay = 0, chain = 0, stage = 0, kw = 0, vs = 0, vw = 0, life = 0, depth = 0;
	for (u64 t = 0; t < tiles; ++t) {
		const u64 *r = &tl[t * 16];
		a += (double)(r[1] - r[0]);
		lay += (double)(r[2] - r[1]);
		chain += (double)(r[3] - r[2]);
		stage += (double)(r[4] - r[2]);
		kw += (double)(r[8] - r[4]);
		vs += (double)(r[9] - r[8]);
		vw += (double)(r[5] - r[9]);
		life += (double)(r[5] - r[0]);
		depth += (double)r[12];
	}
	const double T = (double)tiles;
	printf("%-44s %.3f ms | cycles per tile: load + count %6.0f | layout %5.0f | rank + stage %6.0f (chain %5.0f inside, depth %.1f) | "
	       "keys out %6.0f | payload staged %6.0f | payload out %6.0f | lifetime %6.0f\n",
	       name, best, a / T, lay / T, stage / T, chain / T, depth / T, kw / T, vs / T, vw / T, life / T);
}

int main(int argc, char **argv)
{
	const int log2n = argc > 1 ? atoi(argv[1]) : 28;
	n = (size_t)1 << log2n;
	CK(hipMalloc(&d_kin, n * 4));
	CK(hipMalloc(&d_kout, n * 4));
	CK(hipMalloc(&d_vin, n * 4));
	CK(hipMalloc(&d_vout, n * 4));
	CK(hipMalloc(&d_hist, 256 * 8));
	CK(hipMalloc(&d_status, 256 + (n / 8192 + 2) * 256 * 4));
	CK(hipMalloc(&d_tl, (n / 8192 + 2) * 16 * 8));
	printf("n = 2^%d pairs, one pass by the keys' low byte (uniform digits); 32 Ki-element tiles, one per CU\n", log2n);
	run<u32, u32>("<u32 keys, u32 payload>", 0);
	run<u32, u32>("<u32 keys, u32 payload>, indices generated", SCATTER_GEN_INDEX);
	run<uint16_t, uint8_t>("<u16 keys -> u8 keys, u32 payload>", 0);
	run<uint8_t, uint8_t>("<u8 keys, u32 payload>, no keys written", SCATTER_SKIP_KEYS);
	return 0;
}
