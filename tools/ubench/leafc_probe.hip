// leafc_probe: rsx_leafc_kernel (csrc/rsx_leafc.hpp), the counting leaves of slots of 8 Ki .. 40 Ki two-byte values, on their own.
// Input: 65536 slots of `cap` two-byte values (what the level-2 pass of a sort without a histogram leaves for 2^log2n u32
// keys), `per` +- 64 random values in each (mode 1: every slot's values in 64 bins of 16; mode 2: one slot in 64 with a few fat
// bins; mode 3: every value of a slot the same), `back` of them in the slot's last places, the leaf table, plan and control block
// of such a sort.  The output is checked on the device: ascending over the whole array (the slots' digits are the upper half),
// key sum and key mix those of the slots.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I radix_sorting_amd/csrc tools/ubench/leafc_probe.hip -o tools/ubench/leafc_probe.bin
// Run:   leafc_probe.bin [log2 n = 31] [mode = 0] [grid = 256]
#include "rsx_scatter2.hpp"
#include "rsx_leafc.hpp"

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
		if (mode == 3)
			v = (blockIdx.x * 40503u) & 0xFFFFu;
		const u32 back = ls.ncols >> 16, front = ls.cnt - back;
		slots[(u64)blockIdx.x * cap + (i < front ? i : cap - LEAF16_BACK + (i - front))] = (uint16_t)v;
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
	const u32 back = ls.ncols >> 16, front = ls.cnt - back;
	u64 x = 0;
	for (u32 i = threadIdx.x; i < ls.cnt; i += blockDim.x) {
		const u32 k = ((u32)blockIdx.x << 16) | slots[(u64)blockIdx.x * cap + (i < front ? i : cap - LEAF16_BACK + (i - front))];
		sum += k;
		x ^= (u64)k * 0x9E3779B97F4A7C15ull;
	}
	atomicAdd((unsigned long long *)&out[1], sum);
	atomicXor((unsigned long long *)&out[2], x);
}

__global__ void diff_kernel(const u32 *a, const u32 *b, u64 n, u64 *out)
{
	u64 bad = 0;
	for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x)
		bad += a[i] != b[i];
	atomicAdd((unsigned long long *)&out[0], bad);
}


int main(int argc, char **argv)
{
	const long long a1 = argc > 1 ? atoll(argv[1]) : 31;   // log2 of the number of keys, or the number itself
	const u32 mode = argc > 2 ? (u32)atoi(argv[2]) : 0;
	const unsigned grid = argc > 3 ? (unsigned)atoi(argv[3]) : 256u;
	const size_t nkeys = a1 <= 40 ? (size_t)1 << a1 : (size_t)a1;
	const u32 per = (u32)(nkeys >> 16), nleaf = 65536;
	const u32 cap = ((per + per / 4 + 255) / 256) * 256;
	uint16_t *d_slots;
	u64 *d_chk;
	Plan *d_plan;
	SegCtl *d_ctl;
	LeafSeg *d_seg;
	u32 *d_out;
	CK(hipMalloc(&d_slots, (size_t)nleaf * cap * 2 + 65536));
	CK(hipMalloc(&d_chk, 24));
	CK(hipMalloc(&d_plan, sizeof(Plan)));
	CK(hipMalloc(&d_ctl, sizeof(SegCtl)));
	CK(hipMalloc(&d_seg, (size_t)nleaf * sizeof(LeafSeg)));
	std::vector<LeafSeg> seg(nleaf);
	u64 acc = 0;
	for (u32 i = 0; i < nleaf; ++i) {
		u32 sz = per - 64 + (u32)((u64)(((i + 1) * 2654435761u) >> 16) * 128 >> 16);
		if (i % 1000 == 7)
			sz = cap;          // a full slot
		if (i % 1000 == 8)
			sz = 1 + i % 13;   // a nearly empty one
		if (i % 1000 == 9)
			sz = 0;
		if (i == nleaf - 1 && acc + sz > 0xFFFFFFFFull - 64)
			sz = 0;
		// (as rsx_pass16a_kernel leaves a slot: whole 64-byte atoms in front, up to LEAF16_BACK values in the last places)
		u32 back = sz < 200 ? 0 : (i * 7u) % (LEAF16_BACK + 1);
		if (i % 3 == 0)
			back = 0;
		if (back) {
			const u32 front = std::min((sz - back) & ~31u, cap - LEAF16_BACK);
			sz = front + back;
		}
		seg[i] = LeafSeg{(u32)acc, sz, 2u | (back << 16), i + 1};
		acc += sz;
	}
	const size_t n = acc;
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
	u64 want[3] = {0, 0, 0}, chk[3];
	CK(hipMemset(d_chk, 0, 24));
	hipLaunchKernelGGL(slot_sum_kernel, dim3(nleaf), dim3(256), 0, 0, (const uint16_t *)d_slots, (const LeafSeg *)d_seg, cap, d_chk);
	CK(hipMemcpy(want, d_chk, 24, hipMemcpyDeviceToHost));
	KdfArgs<u32> ka{0, 0, 0};
	auto run = [&](const char *name, auto cfg, unsigned g) {
		typedef decltype(cfg) C;
		if (cap > (u32)C::CAP)
			return;
		float best = 1e9f;
		for (int rep = 0; rep < 4; ++rep) {
			CK(hipMemset(d_out, 0xEE, n * 4));
			hipEvent_t e0, e1;
			CK(hipEventCreate(&e0));
			CK(hipEventCreate(&e1));
			CK(hipEventRecord(e0, 0));
			hipLaunchKernelGGL((rsx_leafc_kernel<u32, C>), dim3(g), dim3(C::BLOCK), 0, 0, d_out, (u32 *)nullptr, (const Plan *)d_plan,
			                   (const LeafSeg *)d_seg, (const SegCtl *)d_ctl, ka, 0u, (u32)C::CAP, (const uint16_t *)d_slots, cap);
			CK(hipEventRecord(e1, 0));
			CK(hipEventSynchronize(e1));
			float ms;
			CK(hipEventElapsedTime(&ms, e0, e1));
			CK(hipGetLastError());
			best = std::min(best, ms);
			CK(hipEventDestroy(e0));
			CK(hipEventDestroy(e1));
		}
		CK(hipMemset(d_chk, 0, 24));
		hipLaunchKernelGGL(check_kernel, dim3(2048), dim3(256), 0, 0, (const u32 *)d_out, (u64)n, d_chk);
		CK(hipMemcpy(chk, d_chk, 24, hipMemcpyDeviceToHost));
		printf("%-34s grid %5u: %.3f ms = %.0f GB/s (6 bytes per key); descents %llu, sum %s, mix %s\n", name, g, best, n * 6.0 / best / 1e6,
		       (unsigned long long)chk[0], chk[1] == want[1] ? "ok" : "DIFFERENT", chk[2] == want[2] ? "ok" : "DIFFERENT");
		fflush(stdout);
	};
	u32 *d_redo;
	CK(hipMalloc(&d_redo, (size_t)nleaf * 4));
	// rsx_leaf16_kernel in shapes for these slots (one leaf per workgroup; what it leaves alone is only counted here)
	auto run16 = [&](const char *name, auto cfg) {
		typedef decltype(cfg) C;
		if (cap > (u32)C::CAP)
			return;
		float best = 1e9f;
		u32 nredo = 0;
		for (int rep = 0; rep < 4; ++rep) {
			CK(hipMemset(d_out, 0xEE, n * 4));
			CK(hipMemset(&d_ctl->nredo, 0, 4));
			hipEvent_t e0, e1;
			CK(hipEventCreate(&e0));
			CK(hipEventCreate(&e1));
			CK(hipEventRecord(e0, 0));
			hipLaunchKernelGGL((rsx_leaf16_kernel<u32, C>), dim3(nleaf), dim3(C::BLOCK), 0, 0, d_out, (u32 *)nullptr, (const Plan *)d_plan,
			                   (const LeafSeg *)d_seg, d_ctl, ka, 0u, (u32)C::CAP, (const uint16_t *)d_slots, cap, d_redo);
			CK(hipEventRecord(e1, 0));
			CK(hipEventSynchronize(e1));
			float ms;
			CK(hipEventElapsedTime(&ms, e0, e1));
			CK(hipGetLastError());
			best = std::min(best, ms);
			CK(hipEventDestroy(e0));
			CK(hipEventDestroy(e1));
			CK(hipMemcpy(&nredo, &d_ctl->nredo, 4, hipMemcpyDeviceToHost));
		}
		CK(hipMemset(d_chk, 0, 24));
		hipLaunchKernelGGL(check_kernel, dim3(2048), dim3(256), 0, 0, (const u32 *)d_out, (u64)n, d_chk);
		CK(hipMemcpy(chk, d_chk, 24, hipMemcpyDeviceToHost));
		printf("%-34s grid %5u: %.3f ms = %.0f GB/s; %u leaves left alone; descents %llu, sum %s, mix %s\n", name, nleaf, best, n * 6.0 / best / 1e6,
		       nredo, (unsigned long long)chk[0], chk[1] == want[1] ? "ok" : "DIFFERENT", chk[2] == want[2] ? "ok" : "DIFFERENT");
		fflush(stdout);
	};
	run16("rsx_leaf16_kernel<256, 6144, 8, 12>", Leaf16Cfg<256, 6144, 8, 12>{});
	run16("rsx_leaf16_kernel<256, 7680, 8, 12>", Leaf16Cfg<256, 7680, 8, 12>{});
	run16("rsx_leaf16_kernel<512, 7680, 8, 12>", Leaf16Cfg<512, 7680, 8, 12>{});
	run16("rsx_leaf16_kernel<512, 7680, 8, 13>", Leaf16Cfg<512, 7680, 8, 13>{});
	run16("rsx_leaf16_kernel<512, 10240, 8, 12>", Leaf16Cfg<512, 10240, 8, 12>{});
	run16("rsx_leaf16_kernel<512, 10240, 8, 13>", Leaf16Cfg<512, 10240, 8, 13>{});
	run16("rsx_leaf16_kernel<512, 15360, 8, 13>", Leaf16Cfg<512, 15360, 8, 13>{});
	run16("rsx_leaf16_kernel<1024, 15360, 8, 14>", Leaf16Cfg<1024, 15360, 8, 14>{});
	run16("rsx_leaf16_kernel<1024, 20480, 8, 13>", Leaf16Cfg<1024, 20480, 8, 13>{});
	run16("rsx_leaf16_kernel<1024, 10240, 8, 13>", Leaf16Cfg<1024, 10240, 8, 13>{});
	run16("rsx_leaf16_kernel<1024, 10240, 8, 14>", Leaf16Cfg<1024, 10240, 8, 14>{});
	run16("rsx_leaf16_kernel<1024, 20480, 8, 14>", Leaf16Cfg<1024, 20480, 8, 14>{});
	run16("rsx_leaf16_kernel<1024, 40960, 4, 14>", Leaf16Cfg<1024, 40960, 4, 14>{});
	run("rsx_leafc_kernel<5 vectors>", LeafCCfg<5>{}, grid);
	run("rsx_leafc_kernel<4 vectors>", LeafCCfg<4>{}, grid);
	run("rsx_leafc_kernel<3 vectors>", LeafCCfg<3>{}, grid);
	run("rsx_leafc_kernel<2 vectors>", LeafCCfg<2>{}, grid);
	run("rsx_leafc_kernel<5 vectors>", LeafCCfg<5>{}, 2 * grid);
	return 0;
}
