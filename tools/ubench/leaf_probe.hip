// leaf_probe: tuning harness for rsx_leaf_sort_kernel (csrc/rsx_hybrid.hpp) on its own.
// Input: 2^log2n u32 keys already grouped by their top 16 bits (what two MSB passes leave), `per` keys per leaf, random low
// bits; a dense leaf table; the device-side plan / control block a two-level sort would have.  Every shape is timed and its
// output checked (sorted + same multiset checksum).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I radix_sorting_amd/csrc tools/ubench/leaf_probe.hip -o tools/ubench/leaf_probe.bin
#include "rsx_scatter2.hpp"

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

__global__ void gen_kernel(u32 *dst, u64 n, u32 per_log2)
{
	for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) {
		u64 z = (i + 1) * 0x9E3779B97F4A7C15ull;
		z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
		z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
		z ^= z >> 31;
		dst[i] = ((u32)(i >> per_log2) << 16) | ((u32)z & 0xFFFFu);
	}
}

// ragged leaves: leaf b holds seg[b].cnt keys with top bits b
__global__ void gen_ragged_kernel(u32 *dst, const LeafSeg *seg)
{
	const LeafSeg ls = seg[blockIdx.x];
	for (u32 i = threadIdx.x; i < ls.cnt; i += blockDim.x) {
		u64 z = ((u64)ls.beg + i + 1) * 0x9E3779B97F4A7C15ull;
		z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
		z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
		z ^= z >> 31;
		dst[ls.beg + i] = ((u32)blockIdx.x << 16) | ((u32)z & 0xFFFFu);
	}
}

__global__ void check_kernel(const u32 *a, u64 n, u64 *out)   // out[0] += descents, out[1] += sum, out[2] ^= xor-ish
{
	u64 bad = 0, sum = 0, x = 0;
	for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) {
		if (i + 1 < n && a[i] > a[i + 1])
			++bad;
		sum += a[i];
		x ^= (u64)a[i] * 0x9E3779B97F4A7C15ull;
	}
	atomicAdd((unsigned long long *)&out[0], bad);
	atomicAdd((unsigned long long *)&out[1], sum);
	atomicXor((unsigned long long *)&out[2], x);
}

static u32 *d_in, *d_out;
static u64 *d_hist, *d_chk;
static Plan *d_plan;
static SegCtl *d_ctl;
static LeafSeg *d_seg;
static size_t n;
static u64 ref_chk[3];

template <typename C>
void bench(const char *name, unsigned grid, u32 lo, u32 hi)
{
	KdfArgs<u32> ka{0, 0, 0};
	// in place on a copy of the input; the second run is the one reported
	float ms = 0;
	for (int rep = 0; rep < 2; ++rep) {
	CK(hipMemcpy(d_out, d_in, n * 4, hipMemcpyDeviceToDevice));
	hipEvent_t e0, e1;
	CK(hipEventCreate(&e0));
	CK(hipEventCreate(&e1));
	CK(hipEventRecord(e0, 0));
	hipLaunchKernelGGL((rsx_leaf_sort_kernel<u32, C>), dim3(grid), dim3(C::BLOCK), 0, 0, d_out, (u32 *)nullptr, (u64)n,
	                   (const u64 *)d_hist, (const Plan *)d_plan, (const LeafSeg *)d_seg, (const SegCtl *)d_ctl, ka,
	                   (u32)HYB_TWO_LEVEL, lo, hi);
	CK(hipEventRecord(e1, 0));
	CK(hipEventSynchronize(e1));
	CK(hipEventElapsedTime(&ms, e0, e1));
	}
	CK(hipMemset(d_chk, 0, 24));
	hipLaunchKernelGGL(check_kernel, dim3(2048), dim3(256), 0, 0, (const u32 *)d_out, (u64)n, d_chk);
	u64 chk[3];
	CK(hipMemcpy(chk, d_chk, 24, hipMemcpyDeviceToHost));
	printf("%-46s grid %6u: %.3f ms = %.0f GB/s; descents %llu, checksum %s\n", name, grid, ms, n * 8.0 / ms / 1e6,
	       (unsigned long long)chk[0], (chk[1] == ref_chk[1] && chk[2] == ref_chk[2]) ? "ok" : "DIFFERENT");
}

int main(int argc, char **argv)
{
	const int log2n = argc > 1 ? atoi(argv[1]) : 28;
	const int per_log2 = argc > 2 ? atoi(argv[2]) : 12;   // keys per leaf
	n = (size_t)1 << log2n;
	const u32 nleaf = (u32)(n >> per_log2), per = 1u << per_log2;
	CK(hipMalloc(&d_in, n * 4));
	CK(hipMalloc(&d_out, n * 4));
	CK(hipMalloc(&d_hist, 8 * 256 * 8));
	CK(hipMalloc(&d_chk, 24));
	CK(hipMalloc(&d_plan, sizeof(Plan)));
	CK(hipMalloc(&d_ctl, sizeof(SegCtl)));
	CK(hipMalloc(&d_seg, (size_t)nleaf * sizeof(LeafSeg)));
	hipLaunchKernelGGL(gen_kernel, dim3(2048), dim3(256), 0, 0, d_in, (u64)n, (u32)per_log2);
	Plan p{};
	p.ncols = 4;
	p.cols[0] = 0, p.cols[1] = 1, p.cols[2] = 2, p.cols[3] = 3;
	p.hyb = HYB_TWO_LEVEL;
	CK(hipMemcpy(d_plan, &p, sizeof p, hipMemcpyHostToDevice));
	SegCtl c{};
	c.mode = SEG_MODE_LEAVES;
	c.maxleaf = per;
	c.nleaf = nleaf;
	CK(hipMemcpy(d_ctl, &c, sizeof c, hipMemcpyHostToDevice));
	const bool ragged = argc > 3 && atoi(argv[3]) != 0;
	std::vector<LeafSeg> seg(nleaf);
	u32 acc = 0, mx = 0;
	for (u32 i = 0; i < nleaf; ++i) {
		// ragged: sizes per +- 64 (what uniform keys give at 4096 per leaf), starts at any element
		u32 sz = ragged ? per - 64 + (((i + 1) * 2654435761u) >> 25) : per;
		if (acc + sz > n)   // (the sizes average a little below `per`: the array's tail stays unused in the ragged case)
			sz = (u32)(n - acc);
		seg[i] = LeafSeg{acc, sz, 2, 0};
		acc += sz;
		mx = std::max(mx, sz);
	}
	CK(hipMemcpy(d_seg, seg.data(), (size_t)nleaf * sizeof(LeafSeg), hipMemcpyHostToDevice));
	if (ragged) {
		n = acc;   // what the leaves cover
		CK(hipMemset(d_in, 0, ((size_t)1 << log2n) * 4));
		hipLaunchKernelGGL(gen_ragged_kernel, dim3(nleaf), dim3(256), 0, 0, d_in, (const LeafSeg *)d_seg);
		c.maxleaf = mx;
		CK(hipMemcpy(d_ctl, &c, sizeof c, hipMemcpyHostToDevice));
		printf("ragged leaves: %u - %u keys, starting at any element\n", per - 64, mx);
		CK(hipMemset(d_chk, 0, 24));
		hipLaunchKernelGGL(check_kernel, dim3(2048), dim3(256), 0, 0, (const u32 *)d_in, (u64)n, d_chk);
		CK(hipMemcpy(ref_chk, d_chk, 24, hipMemcpyDeviceToHost));
	}
	CK(hipMemset(d_chk, 0, 24));
	hipLaunchKernelGGL(check_kernel, dim3(2048), dim3(256), 0, 0, (const u32 *)d_in, (u64)n, d_chk);
	CK(hipMemcpy(ref_chk, d_chk, 24, hipMemcpyDeviceToHost));
	printf("n = 2^%d u32 keys in %u leaves of %u keys, two columns per leaf\n", log2n, nleaf, per);
	{
		// the floor: a copy of the same bytes
		hipEvent_t e0, e1;
		CK(hipEventCreate(&e0));
		CK(hipEventCreate(&e1));
		CK(hipEventRecord(e0, 0));
		CK(hipMemcpyAsync(d_out, d_in, n * 4, hipMemcpyDeviceToDevice, 0));
		CK(hipEventRecord(e1, 0));
		CK(hipEventSynchronize(e1));
		float ms;
		CK(hipEventElapsedTime(&ms, e0, e1));
		printf("device-to-device copy of the keys: %.3f ms\n", ms);
	}
#define SHAPE(NW, KPT, WPE, RANK1, PRE, GRID) \
	bench<LeafCfg<u32, NW, KPT, WPE, RANK1, PRE>>("NW " #NW " KPT " #KPT " WPE " #WPE " rank1 " #RANK1 " prefetch " #PRE, GRID, 0u, 1u << 20)
	SHAPE(4, 32, 4, true, true, 1024);
	SHAPE(4, 32, 4, false, false, nleaf);
	SHAPE(4, 32, 4, true, false, nleaf);
	SHAPE(4, 32, 4, true, false, 1024);
	SHAPE(4, 32, 4, true, false, 2048);
	SHAPE(4, 32, 4, true, false, 8192);
	SHAPE(4, 32, 2, true, false, nleaf);
	SHAPE(8, 16, 8, true, false, nleaf);
	SHAPE(8, 16, 4, true, false, nleaf);
	SHAPE(8, 32, 4, true, false, nleaf);
	// shapes that hold just the 5120 keys a slot of 2^28 keys can (more workgroups per CU: 24.5 KiB of LDS each)
	SHAPE(4, 20, 4, true, false, 8192);
	SHAPE(4, 20, 5, true, false, 8192);
	SHAPE(4, 20, 6, true, false, 8192);
	SHAPE(4, 20, 6, true, false, 16384);
	SHAPE(4, 24, 5, true, false, 8192);
	SHAPE(8, 12, 8, true, false, 8192);
	SHAPE(8, 12, 6, true, false, 8192);
	SHAPE(6, 16, 6, true, false, 8192);
	SHAPE(6, 16, 4, true, false, 8192);
	SHAPE(4, 20, 4, true, false, 4096);
	SHAPE(4, 20, 4, true, false, 65536);
	// small leaves (2^26 keys in 65536 leaves of 1024: leaf_probe 26 10 1)
	SHAPE(4, 8, 4, true, false, 65536);
	SHAPE(4, 8, 8, true, false, 65536);
	SHAPE(4, 12, 6, true, false, 65536);
	return 0;
}
