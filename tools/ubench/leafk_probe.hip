// leafk_probe: the leaves of 8-byte keys (rsx_leafk_kernel<u64, u64>, csrc/rsx_leaf16.hpp) against the variants of
// tools/ubench/rsx_leafk2.hpp on the same slots.
// Input: 65536 slots of `cap` whole u64 keys (what the level-2 pass of a sort without a histogram leaves for 2^log2n uniform
// u64 keys: slot s holds keys whose top sixteen bits are s), `per` +- 64 random keys in each, with the leaf table, plan and
// control block of such a sort.  Every variant is timed and its output compared element for element with the shipped
// kernel's, which is checked for sortedness and checksum.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I radix_sorting_amd/csrc tools/ubench/leafk_probe.hip -o tools/ubench/leafk_probe.bin
#include "rsx_scatter2.hpp"
#include "rsx_leaf16.hpp"
#include "rsx_leafk2.hpp"

#include <algorithm>
#include <functional>
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

__global__ void gen_slots_kernel(u64 *slots, const LeafSeg *seg, u32 cap, u32 mode)
{
	const LeafSeg ls = seg[blockIdx.x];
	for (u32 i = threadIdx.x; i < ls.cnt; i += blockDim.x) {
		u64 z = ((u64)ls.beg + i + 1) * 0x9E3779B97F4A7C15ull;
		z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
		z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
		z ^= z >> 31;
		u64 v = z & 0xFFFFFFFFFFFFull;
		if (mode == 1)
			v &= 0xFC0FFFFFFFFFull;   // 64 bins of the top twelve bits: bins of ~64 keys (every leaf goes to the list)
		if (mode == 2 && (blockIdx.x & 63u) == 0)
			v &= 0xFFF0FFFFFFFFull | ((v >> 44) << 32);   // one slot in 64 with a few fat bins
		if (mode == 3)
			v &= 0xFFFFFFFF0000ull;   // ties in the low sixteen bits
		slots[(u64)blockIdx.x * cap + i] = ((u64)blockIdx.x << 48) | v;
	}
}

__global__ void check_kernel(const u64 *a, u64 n, u64 *out)
{
	u64 bad = 0, sum = 0, x = 0;
	for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) {
		if (i + 1 < n && a[i] > a[i + 1])
			++bad;
		sum += a[i];
		x ^= a[i] * 0x9E3779B97F4A7C15ull;
	}
	atomicAdd((unsigned long long *)&out[0], bad);
	atomicAdd((unsigned long long *)&out[1], sum);
	atomicXor((unsigned long long *)&out[2], x);
}

__global__ void slot_sum_kernel(const u64 *slots, const LeafSeg *seg, u32 cap, u64 *out)
{
	const LeafSeg ls = seg[blockIdx.x];
	u64 sum = 0;
	for (u32 i = threadIdx.x; i < ls.cnt; i += blockDim.x)
		sum += slots[(u64)blockIdx.x * cap + i];
	atomicAdd((unsigned long long *)&out[1], sum);
}

__global__ void diff_kernel(const u64 *a, const u64 *b, u64 n, u64 *out)
{
	u64 bad = 0;
	for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x)
		bad += a[i] != b[i];
	atomicAdd((unsigned long long *)&out[0], bad);
}

static u64 *d_ref, *d_out, *d_slots;
static u64 *d_chk;
static Plan *d_plan;
static SegCtl *d_ctl;
static LeafSeg *d_seg;
static u32 *d_redo;
static size_t n;
static u32 cap;
static u32 nleaf = 65536;

static float timed(const std::function<void()> &f)
{
	hipEvent_t e0, e1;
	CK(hipEventCreate(&e0));
	CK(hipEventCreate(&e1));
	CK(hipEventRecord(e0, 0));
	f();
	CK(hipEventRecord(e1, 0));
	CK(hipEventSynchronize(e1));
	float ms;
	CK(hipEventElapsedTime(&ms, e0, e1));
	CK(hipGetLastError());
	CK(hipEventDestroy(e0));
	CK(hipEventDestroy(e1));
	return ms;
}

// the list launch behind a leafk launch (rsx.hip, launch_leaves): the leaves the register passes handed on
static float run_list(u64 *out)
{
	typedef LeafCfg<u64, 4, 32, 2, true, false> S;
	KdfArgs<u64> ka{0, 0, 0};
	return timed([&] {
		hipLaunchKernelGGL((rsx_leaf_sort_kernel<u64, S>), dim3(2048), dim3(S::BLOCK), 0, 0, out, (u64 *)nullptr, (u64)n,
		                   (const u64 *)nullptr, (const Plan *)d_plan, (const LeafSeg *)d_seg, (const SegCtl *)d_ctl, ka,
		                   (u32)HYB_TWO_LEVEL, 0u, (u32)S::CAP, (const u64 *)d_slots, cap, 1u, (const u64 *)nullptr,
		                   (const u32 *)d_redo);
	});
}

template <typename F> void bench(const char *name, unsigned grid, bool is_ref, F &&launch)
{
	float ms = 0, best = 1e9f, ms_list = 0;
	u32 nredo = 0;
	u64 *out = is_ref ? d_ref : d_out;
	for (int rep = 0; rep < 4; ++rep) {
		CK(hipMemset(out, 0xEE, n * 8));
		CK(hipMemset(&d_ctl->nredo, 0, 4));
		ms = timed([&] { launch(out, grid); });
		best = std::min(best, ms);
		ms_list = run_list(out);
		CK(hipMemcpy(&nredo, &d_ctl->nredo, 4, hipMemcpyDeviceToHost));
	}
	u64 chk[3];
	CK(hipMemset(d_chk, 0, 24));
	if (is_ref) {
		u64 want[3] = {0, 0, 0};
		hipLaunchKernelGGL(slot_sum_kernel, dim3(nleaf), dim3(256), 0, 0, (const u64 *)d_slots, (const LeafSeg *)d_seg, cap, d_chk);
		CK(hipMemcpy(want, d_chk, 24, hipMemcpyDeviceToHost));
		CK(hipMemset(d_chk, 0, 24));
		hipLaunchKernelGGL(check_kernel, dim3(2048), dim3(256), 0, 0, (const u64 *)d_ref, (u64)n, d_chk);
		CK(hipMemcpy(chk, d_chk, 24, hipMemcpyDeviceToHost));
		printf("%-54s grid %6u: %.3f (best %.3f) ms + %.3f for %u listed = %.0f GB/s; descents %llu, sum %s\n", name, grid, ms, best,
		       ms_list, nredo, n * 16.0 / best / 1e6, (unsigned long long)chk[0], chk[1] == want[1] ? "ok" : "DIFFERENT");
	} else {
		hipLaunchKernelGGL(diff_kernel, dim3(2048), dim3(256), 0, 0, (const u64 *)d_out, (const u64 *)d_ref, (u64)n, d_chk);
		CK(hipMemcpy(chk, d_chk, 24, hipMemcpyDeviceToHost));
		printf("%-54s grid %6u: %.3f (best %.3f) ms + %.3f for %u listed = %.0f GB/s; differences %llu\n", name, grid, ms, best,
		       ms_list, nredo, n * 16.0 / best / 1e6, (unsigned long long)chk[0]);
	}
	fflush(stdout);
}

template <typename C> void launch_old(u64 *out, unsigned grid)
{
	KdfArgs<u64> ka{0, 0, 0};
	hipLaunchKernelGGL((rsx_leafk_kernel<u64, u64, C>), dim3(grid), dim3(C::BLOCK), 0, 0, out, (u64 *)nullptr, (const Plan *)d_plan,
	                   (const LeafSeg *)d_seg, d_ctl, ka, 0u, (u32)C::CAP, (const u64 *)d_slots, cap, d_redo, 25u);
}
static void launch_k8(u64 *out, unsigned grid)
{
	KdfArgs<u64> ka{0, 0, 0};
	hipLaunchKernelGGL((rsx_leafk8_kernel<u64, u64, LeafK8Cfg>), dim3(grid), dim3(LeafK8Cfg::BLOCK), 0, 0, out, (u64 *)nullptr,
	                   (const Plan *)d_plan, (const LeafSeg *)d_seg, d_ctl, ka, 0u, (u32)LeafK8Cfg::CAP, (const u64 *)d_slots, cap,
	                   d_redo, 25u);
}
template <typename C> void launch_new(u64 *out, unsigned grid)
{
	KdfArgs<u64> ka{0, 0, 0};
	hipLaunchKernelGGL((rsx_leafk2_kernel<u64, u64, C>), dim3(grid), dim3(C::BLOCK), 0, 0, out, (u64 *)nullptr, (const Plan *)d_plan,
	                   (const LeafSeg *)d_seg, d_ctl, ka, 0u, (u32)C::CAP, (const u64 *)d_slots, cap, d_redo, 25u);
}

int main(int argc, char **argv)
{
	const long long a1 = argc > 1 ? atoll(argv[1]) : 28;
	const size_t nkeys = a1 <= 40 ? (size_t)1 << a1 : (size_t)a1;
	const u32 mode = argc > 2 ? (u32)atoi(argv[2]) : 0;
	const u32 per = (u32)(nkeys >> 16);
	if (argc > 3)
		nleaf = 1u << atoi(argv[3]);
	cap = ((per + per / 4 + 255) / 256) * 256;
	CK(hipMalloc(&d_slots, (size_t)nleaf * cap * 8 + 65536));
	CK(hipMalloc(&d_chk, 24));
	CK(hipMalloc(&d_plan, sizeof(Plan)));
	CK(hipMalloc(&d_ctl, sizeof(SegCtl)));
	CK(hipMalloc(&d_seg, (size_t)nleaf * sizeof(LeafSeg)));
	CK(hipMalloc(&d_redo, (size_t)nleaf * 4));
	std::vector<LeafSeg> seg(nleaf);
	u32 acc = 0, mx = 0;
	for (u32 i = 0; i < nleaf; ++i) {
		const u32 spread = per >= 1024 ? 64 : per / 16;
		u32 sz = per - spread + (u32)((u64)(((i + 1) * 2654435761u) >> 16) * (2 * spread) >> 16);
		if (i % 1000 == 7)
			sz = cap;
		if (i % 1000 == 8)
			sz = 1 + i % 13;
		if (i % 1000 == 9)
			sz = 0;
		seg[i] = LeafSeg{acc, sz, 6, i + 1};
		acc += sz;
		mx = std::max(mx, sz);
	}
	n = acc;
	CK(hipMalloc(&d_ref, n * 8 + 64));
	CK(hipMalloc(&d_out, n * 8 + 64));
	CK(hipMemcpy(d_seg, seg.data(), (size_t)nleaf * sizeof(LeafSeg), hipMemcpyHostToDevice));
	hipLaunchKernelGGL(gen_slots_kernel, dim3(nleaf), dim3(256), 0, 0, d_slots, (const LeafSeg *)d_seg, cap, mode);
	Plan p{};
	p.ncols = 8;
	for (int i = 0; i < 8; ++i)
		p.cols[i] = i;
	p.hyb = HYB_TWO_LEVEL;
	CK(hipMemcpy(d_plan, &p, sizeof p, hipMemcpyHostToDevice));
	SegCtl c{};
	c.mode = SEG_MODE_LEAVES;
	c.maxleaf = cap;
	c.nleaf = nleaf;
	c.leaf16 = 1;
	c.shift1 = 56;
	c.shift2 = 48;
	CK(hipMemcpy(d_ctl, &c, sizeof c, hipMemcpyHostToDevice));
	printf("n = %zu u64 keys in %u slots of %u keys (%u +- 64 in each), mode %u\n", n, nleaf, cap, per, mode);
	typedef LeafKCfg<512, 5120, 8, 12> K8;
	bench("rsx_leafk_kernel<u64, u64, <512, 5120, 8, 12>> (csrc, general)", nleaf, true, [&](u64 *o, unsigned g) { launch_old<K8>(o, g); });
	bench("rsx_leafk8_kernel (csrc: 6-byte staging, 16-byte loads)", nleaf, false, [&](u64 *o, unsigned g) { launch_k8(o, g); });
#define NEW(GRID, ...) bench("leafk2 <" #__VA_ARGS__ ">", GRID, false, [&](u64 *o, unsigned g) { launch_new<LeafK2Cfg<__VA_ARGS__>>(o, g); })
	NEW(nleaf, 512, 5120, 6, 12, false);
	NEW(nleaf, 512, 5120, 6, 12, true);
	NEW(nleaf, 512, 5120, 6, 12, false, 0, true);
	NEW(nleaf, 512, 5120, 8, 12, false, 0, true);
	NEW(nleaf, 512, 5120, 8, 12, false, 0, true, true);
	NEW(nleaf, 512, 5120, 8, 12, false, 7, true, true);
	NEW(nleaf, 512, 5120, 8, 12, false, 7, true);
	NEW(nleaf, 512, 5120, 6, 12, false, 1);
	NEW(nleaf, 512, 5120, 6, 12, false, 2);
	NEW(nleaf, 512, 5120, 6, 12, false, 3);
	NEW(nleaf, 512, 5120, 6, 12, false, 4);
	NEW(nleaf, 512, 5120, 6, 12, false, 5);
	NEW(nleaf, 512, 5120, 6, 12, false, 7);
	return 0;
}
