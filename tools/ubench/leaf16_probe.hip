// leaf16_probe: rsx_leaf16_kernel (csrc/rsx_leaf16.hpp) against rsx_leaf_sort_kernel<..., u16, DENSE> on the same slots.
// Input: 65536 slots of `cap` two-byte values (what the level-2 pass of a sort without a histogram leaves for 2^log2n u32
// keys), `per` +- 64 random values in each (or a clustered pattern: mode 1 = every slot's values in 64 bins of 16), the
// leaf table, plan and control block of such a sort.  Every variant is timed, its output compared element for element with
// the reference kernel's, which is checked for sortedness and checksum.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I radix_sorting_amd/csrc tools/ubench/leaf16_probe.hip -o tools/ubench/leaf16_probe.bin
#include "rsx_scatter2.hpp"
#include "rsx_leaf16.hpp"

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

__global__ void gen_slots_kernel(uint16_t *slots, const LeafSeg *seg, u32 cap, u32 mode)
{
	const LeafSeg ls = seg[blockIdx.x];
	for (u32 i = threadIdx.x; i < ls.cnt; i += blockDim.x) {
		u64 z = ((u64)ls.beg + i + 1) * 0x9E3779B97F4A7C15ull;
		z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
		z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
		z ^= z >> 31;
		u32 v = (u32)z & 0xFFFFu;
		if (mode == 1)
			v &= 0xFC0Fu;   // 64 bins of the top twelve bits, 16 values each: bins of ~64 keys
		if (mode == 2 && (blockIdx.x & 63u) == 0)
			v &= 0xFFF0u | (v >> 12);   // one slot in 64 with a few fat bins
		slots[(u64)blockIdx.x * cap + i] = (uint16_t)v;
	}
}

__global__ void check_kernel(const u32 *a, u64 n, u64 *out)
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

__global__ void slot_sum_kernel(const uint16_t *slots, const LeafSeg *seg, u32 cap, u64 *out)
{
	const LeafSeg ls = seg[blockIdx.x];
	u64 sum = 0;
	for (u32 i = threadIdx.x; i < ls.cnt; i += blockDim.x)
		sum += ((u32)blockIdx.x << 16) | slots[(u64)blockIdx.x * cap + i];
	atomicAdd((unsigned long long *)&out[1], sum);
}

__global__ void diff_kernel(const u32 *a, const u32 *b, u64 n, u64 *out)
{
	u64 bad = 0;
	for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x)
		bad += a[i] != b[i];
	atomicAdd((unsigned long long *)&out[0], bad);
}

static u32 *d_ref, *d_out;
static uint16_t *d_slots;
static u64 *d_chk;
static Plan *d_plan;
static SegCtl *d_ctl;
static LeafSeg *d_seg;
static u32 *d_redo;
static size_t n;
static u32 cap;
static u32 nleaf = 65536;   // (argv[3]: 2^k leaves only -- do slots that are still in the Infinity Cache read faster?)

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
	return ms;
}

template <typename OldCfg> float run_old(u32 *out, unsigned grid, const u32 *redo = nullptr)
{
	KdfArgs<u32> ka{0, 0, 0};
	return timed([&] {
		hipLaunchKernelGGL((rsx_leaf_sort_kernel<u32, OldCfg, uint16_t, true>), dim3(grid), dim3(OldCfg::BLOCK), 0, 0, out,
		                   (u32 *)nullptr, (u64)n, (const u64 *)nullptr, (const Plan *)d_plan, (const LeafSeg *)d_seg,
		                   (const SegCtl *)d_ctl, ka, (u32)HYB_TWO_LEVEL, 0u, (u32)OldCfg::CAP, (const u32 *)d_slots, cap, 0u,
		                   (const u64 *)nullptr, redo);
	});
}

template <typename C, typename OldCfg> void bench_new(const char *name, unsigned grid)
{
	KdfArgs<u32> ka{0, 0, 0};
	float ms = 0, ms_redo = 0;
	u32 nredo = 0;
	for (int rep = 0; rep < 3; ++rep) {
		CK(hipMemset(d_out, 0xEE, n * 4));
		CK(hipMemset(&d_ctl->nredo, 0, 4));
		ms = timed([&] {
			hipLaunchKernelGGL((rsx_leaf16_kernel<u32, C>), dim3(grid), dim3(C::BLOCK), 0, 0, d_out, (u32 *)nullptr,
			                   (const Plan *)d_plan, (const LeafSeg *)d_seg, d_ctl, ka, 0u, (u32)C::CAP,
			                   (const uint16_t *)d_slots, cap, d_redo);
		});
		ms_redo = run_old<OldCfg>(d_out, 1024, d_redo);
		CK(hipMemcpy(&nredo, &d_ctl->nredo, 4, hipMemcpyDeviceToHost));
	}
	CK(hipMemset(d_chk, 0, 24));
	hipLaunchKernelGGL(diff_kernel, dim3(2048), dim3(256), 0, 0, (const u32 *)d_out, (const u32 *)d_ref, (u64)n, d_chk);
	u64 chk[3];
	CK(hipMemcpy(chk, d_chk, 24, hipMemcpyDeviceToHost));
	printf("%-40s grid %6u: %.3f ms + %.3f ms for %u leaves left over = %.0f GB/s; differences %llu\n", name, grid, ms, ms_redo,
	       nredo, n * 6.0 / (ms + ms_redo) / 1e6, (unsigned long long)chk[0]);
}

template <typename C, typename OldCfg> void bench_wave(const char *name, unsigned grid)
{
	KdfArgs<u32> ka{0, 0, 0};
	float ms = 0, ms_redo = 0;
	u32 nredo = 0;
	for (int rep = 0; rep < 3; ++rep) {
		CK(hipMemset(d_out, 0xEE, n * 4));
		CK(hipMemset(&d_ctl->nredo, 0, 4));
		ms = timed([&] {
			hipLaunchKernelGGL((rsx_leaf16w_kernel<u32, C>), dim3(grid), dim3(C::BLOCK), 0, 0, d_out, (u32 *)nullptr,
			                   (const Plan *)d_plan, (const LeafSeg *)d_seg, d_ctl, ka, 0u, (u32)C::CAP,
			                   (const uint16_t *)d_slots, cap);
		});
		CK(hipMemcpy(&nredo, &d_ctl->nredo, 4, hipMemcpyDeviceToHost));
	}
	CK(hipMemset(d_chk, 0, 24));
	hipLaunchKernelGGL(diff_kernel, dim3(2048), dim3(256), 0, 0, (const u32 *)d_out, (const u32 *)d_ref, (u64)n, d_chk);
	u64 chk[3];
	CK(hipMemcpy(chk, d_chk, 24, hipMemcpyDeviceToHost));
	printf("%-40s grid %6u: %.3f ms + %.3f ms for %u leaves left over = %.0f GB/s; differences %llu\n", name, grid, ms, ms_redo,
	       nredo, n * 6.0 / (ms + ms_redo) / 1e6, (unsigned long long)chk[0]);
}

int main(int argc, char **argv)
{
	// argv[1]: log2 of the number of keys, or the number itself (4 * 10^7: the reference's headline)
	const long long a1 = argc > 1 ? atoll(argv[1]) : 28;
	const size_t nkeys = a1 <= 40 ? (size_t)1 << a1 : (size_t)a1;
	const u32 mode = argc > 2 ? (u32)atoi(argv[2]) : 0;
	const u32 per = (u32)(nkeys >> 16);
	if (argc > 3)
		nleaf = 1u << atoi(argv[3]);
	cap = ((per + per / 4 + 255) / 256) * 256;
	CK(hipMalloc(&d_slots, (size_t)nleaf * cap * 2 + 65536));
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
			sz = cap;          // a full slot
		if (i % 1000 == 8)
			sz = 1 + i % 13;   // a nearly empty one
		if (i % 1000 == 9)
			sz = 0;
		seg[i] = LeafSeg{acc, sz, 2, i + 1};
		acc += sz;
		mx = std::max(mx, sz);
	}
	n = acc;
	CK(hipMalloc(&d_ref, n * 4 + 64));
	CK(hipMalloc(&d_out, n * 4 + 64));
	CK(hipMemcpy(d_seg, seg.data(), (size_t)nleaf * sizeof(LeafSeg), hipMemcpyHostToDevice));
	hipLaunchKernelGGL(gen_slots_kernel, dim3(nleaf), dim3(256), 0, 0, d_slots, (const LeafSeg *)d_seg, cap, mode);
	Plan p{};
	p.ncols = 4;
	p.cols[0] = 0, p.cols[1] = 1, p.cols[2] = 2, p.cols[3] = 3;
	p.hyb = HYB_TWO_LEVEL;
	CK(hipMemcpy(d_plan, &p, sizeof p, hipMemcpyHostToDevice));
	SegCtl c{};
	c.mode = SEG_MODE_LEAVES;
	c.maxleaf = cap;
	c.nleaf = nleaf;
	c.leaf16 = 1;
	c.shift1 = 24;
	c.shift2 = 16;
	CK(hipMemcpy(d_ctl, &c, sizeof c, hipMemcpyHostToDevice));
	printf("n = %zu u32 keys in %u slots of %u two-byte values (%u +- 64 in each), mode %u\n", n, nleaf, cap, per, mode);
	typedef LeafCfg<u32, 4, 20, 4, true, false> Fit;
	float ms = 0;
	for (int rep = 0; rep < 3; ++rep)
		ms = run_old<Fit>(d_ref, nleaf);
	u64 want[3] = {0, 0, 0}, chk[3];
	CK(hipMemset(d_chk, 0, 24));
	hipLaunchKernelGGL(slot_sum_kernel, dim3(nleaf), dim3(256), 0, 0, (const uint16_t *)d_slots, (const LeafSeg *)d_seg, cap, d_chk);
	CK(hipMemcpy(want, d_chk, 24, hipMemcpyDeviceToHost));
	CK(hipMemset(d_chk, 0, 24));
	hipLaunchKernelGGL(check_kernel, dim3(2048), dim3(256), 0, 0, (const u32 *)d_ref, (u64)n, d_chk);
	CK(hipMemcpy(chk, d_chk, 24, hipMemcpyDeviceToHost));
	printf("%-40s grid %6u: %.3f ms = %.0f GB/s; descents %llu, sum %s\n", "rsx_leaf_sort_kernel<u32, 4 x 20, u16, DENSE>", nleaf, ms,
	       n * 6.0 / ms / 1e6, (unsigned long long)chk[0], chk[1] == want[1] ? "ok" : "DIFFERENT");
#define NEW(BLK, CAPV, WPE, NB, GRID) bench_new<Leaf16Cfg<BLK, CAPV, WPE, NB>, Fit>("rsx_leaf16_kernel<" #BLK ", " #CAPV ", " #WPE ", " #NB ">", GRID)
#define WAVE(CAPV, NB, NWV, GRID) bench_wave<Leaf16WCfg<CAPV, NB, NWV>, Fit>("rsx_leaf16w_kernel<" #CAPV ", " #NB ", " #NWV ">", GRID)
	if (cap <= 1024) {
		WAVE(1024, 10, 4, nleaf / 4);
		WAVE(1024, 10, 8, nleaf / 8);
		WAVE(1024, 10, 4, 4096);
		WAVE(1024, 10, 4, 2048);
		if (cap <= 512) {
			WAVE(512, 9, 4, nleaf / 4);
			WAVE(512, 9, 8, nleaf / 8);
		}
	}
	if (cap <= 2048) {
		WAVE(2048, 10, 4, nleaf / 4);
		WAVE(2048, 10, 2, nleaf / 2);
		WAVE(2048, 10, 8, nleaf / 8);
	}
	if (cap <= 1536) {
		NEW(128, 1536, 8, 11, nleaf);
		NEW(128, 1536, 8, 10, nleaf);
	}
	if (cap <= 2048) {
		NEW(128, 2048, 8, 11, nleaf);
		NEW(256, 2048, 8, 11, nleaf);
	}
	if (cap <= 2560) {
		NEW(128, 2560, 8, 11, nleaf);
		NEW(256, 2560, 8, 11, nleaf);
		NEW(256, 2560, 8, 11, 8192);
	}
	if (cap > 2560 && cap <= 3840) {
		NEW(256, 3840, 8, 12, nleaf);
		NEW(128, 3840, 8, 11, nleaf);
		NEW(256, 3840, 8, 11, nleaf);
	}
	if (cap > 2560) {
		NEW(128, 5120, 8, 11, nleaf);
		NEW(256, 5120, 8, 11, nleaf);
	}
	NEW(256, 5120, 8, 12, nleaf);
	if (cap > 2560) {
		NEW(256, 5120, 8, 12, 8192);
#define SKIPV(SK) bench_new<Leaf16Cfg<256, 5120, 8, 12, SK>, Fit>("  probe: skip mask " #SK " (wrong output)", nleaf)
		SKIPV(1);
		SKIPV(2);
		SKIPV(3);
		SKIPV(4);
		SKIPV(5);
		SKIPV(7);
	}
	return 0;
}
