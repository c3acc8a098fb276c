// pass32_probe: the level-1 pass of a keys-only sort without a histogram on its own (csrc/rsx_pass32.hpp: whole 64-byte atoms,
// cursors, carried keys) -- by the keys' top byte, and by 255 SPLITTERS searched in the LDS (DIG == 2): what a sample-sort
// style level-1 pass would cost against the byte pass (round-4 review, item 2b).  With splitters i << 24 the two make the same
// partition, so one check serves both: every key of slot d has top byte d, the slots hold n keys, their key sum and key mix are
// the input's.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I radix_sorting_amd/csrc -I tools/ubench tools/ubench/pass32_probe.hip -o tools/ubench/pass32_probe.bin
// Run:   pass32_probe.bin [log2 n = 28] [mode: 0 uniform keys | 1 Zipf-like 32-bit keys with quantile splitters] [key bytes: 4 | 8] [1 | 0: the rank kept from the counting atomic | a second atomic]
#include "rsx_scatter2.hpp"
#include "rsx_pass32_splitters.hpp"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

using namespace rsx;

#define CK(x)                                                                             \
	do {                                                                                  \
		hipError_t e_ = (x);                                                              \
		if (e_ != hipSuccess) {                                                           \
			printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
			exit(1);                                                                      \
		}                                                                                 \
	} while (0)

template <typename KT> __global__ void gen_kernel(KT *k, u64 n, u32 mode)
{
	for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) {
		u64 z = (i + 1) * 0x9E3779B97F4A7C15ull;
		z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
		z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
		z ^= z >> 31;
		KT v = (KT)z;
		if (mode == 1 && sizeof(KT) == 4) {   // log-uniform over [2^8, 2^32): b = 9 .. 32 uniformly, then b - 1 random bits (SURVEY.md 8d cfg 3 (iv), in 32 bits)
			const u32 b = 9 + (u32)((z >> 58) % 24);
			v = (KT)((1u << (b - 1)) | ((u32)z & ((1u << (b - 1)) - 1u)));
		}
		k[i] = v;
	}
}

// every key of [slot d's front and back] lies between the slot's splitters; sums over all slots
template <typename KT> __global__ void check_kernel(const KT *slots, const u32 *cursors, u32 cap, u32 back_cap, const KT *spl, u64 *out)
{
	const u32 d = blockIdx.x;
	const u32 front = cursors[d], back = cursors[256 + d];
	const KT lo = d ? spl[d - 1] : (KT)0, hi = d < 255 ? spl[d] : (KT)~(KT)0;
	u64 bad = 0, sum = 0, mix = 0;
	for (u32 i = threadIdx.x; i < front + back; i += blockDim.x) {
		const KT k = i < front ? slots[(u64)d * cap + i] : slots[(u64)d * cap + cap - back_cap + (i - front)];
		bad += (k < lo || (d < 255 && k >= hi)) ? 1 : 0;
		sum += (u64)k;
		mix ^= (u64)k * 0x9E3779B97F4A7C15ull;
	}
	atomicAdd((unsigned long long *)&out[0], bad);
	atomicAdd((unsigned long long *)&out[1], sum);
	atomicXor((unsigned long long *)&out[2], mix);
	if (threadIdx.x == 0)
		atomicAdd((unsigned long long *)&out[3], (u64)front + back);
}

template <typename KT> __global__ void sum_kernel(const KT *k, u64 n, u64 *out)
{
	u64 sum = 0, mix = 0;
	for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) {
		sum += (u64)k[i];
		mix ^= (u64)k[i] * 0x9E3779B97F4A7C15ull;
	}
	atomicAdd((unsigned long long *)&out[1], sum);
	atomicXor((unsigned long long *)&out[2], mix);
}

template <typename KT, typename CFG> int run_all(int log2n, u32 mode)
{
	const size_t n = (size_t)1 << log2n;
	KT *d_in, *d_slots, *d_spl;
	u32 *d_cur, *d_ovf;
	u64 *d_chk;
	SegCtl *d_ctl;
	CK(hipMalloc(&d_in, n * sizeof(KT)));
	hipLaunchKernelGGL((gen_kernel<KT>), dim3(2048), dim3(256), 0, 0, d_in, (u64)n, mode);
	// splitters: the byte boundaries (mode 0), or the 255 quantiles of a sorted sample of 64 Ki keys (mode 1)
	std::vector<KT> spl(255);
	if (mode == 0) {
		for (u32 i = 0; i < 255; ++i)
			spl[i] = (KT)(i + 1) << (8 * sizeof(KT) - 8);
	} else {
		std::vector<KT> sample(65536);
		for (u32 i = 0; i < 65536; ++i)
			CK(hipMemcpy(&sample[i], d_in + (size_t)i * (n / 65536), sizeof(KT), hipMemcpyDeviceToHost));
		std::sort(sample.begin(), sample.end());
		for (u32 i = 0; i < 255; ++i)
			spl[i] = sample[(i + 1) * 256];
	}
	const u32 mean = (u32)(n >> 8);
	const u32 cap = ((mode == 0 ? mean + mean / 4 : 2 * mean) + 255) / 256 * 256;   // (quantile splitters from a sample: wider slots)
	CK(hipMalloc(&d_slots, ((size_t)256 * cap + 65536) * sizeof(KT)));
	CK(hipMalloc(&d_cur, 4096));
	CK(hipMalloc(&d_ovf, 64));
	CK(hipMalloc(&d_spl, 2048));
	CK(hipMalloc(&d_chk, 64));
	CK(hipMalloc(&d_ctl, sizeof(SegCtl)));
	CK(hipMemcpy(d_spl, spl.data(), 255 * sizeof(KT), hipMemcpyHostToDevice));
	SegCtl c{};
	c.blind = BLIND_GO;
	c.shift1 = 8 * sizeof(KT) - 8;
	CK(hipMemcpy(d_ctl, &c, sizeof c, hipMemcpyHostToDevice));
	KdfArgs<KT> ka{0, 0, 0};
	u64 want[4] = {0, 0, 0, 0};
	CK(hipMemset(d_chk, 0, 64));
	hipLaunchKernelGGL((sum_kernel<KT>), dim3(2048), dim3(256), 0, 0, (const KT *)d_in, (u64)n, d_chk);
	CK(hipMemcpy(want, d_chk, 32, hipMemcpyDeviceToHost));
	printf("n = %zu %zu-byte keys, mode %u, slots of %u keys\n", n, sizeof(KT), mode, cap);
	auto run = [&](const char *name, int which) {
		float best = 1e9f;
		for (int rep = 0; rep < 5; ++rep) {
			CK(hipMemset(d_cur, 0, 4096));
			CK(hipMemset(d_ovf, 0, 64));
			hipEvent_t e0, e1;
			CK(hipEventCreate(&e0));
			CK(hipEventCreate(&e1));
			CK(hipEventRecord(e0, 0));
			if (which == 0)
				hipLaunchKernelGGL((rsx_pass32s_kernel<KT, 1, true, CFG>), dim3(256), dim3(1024), 0, 0, (const KT *)d_in, (u64)n, d_slots, 0u, 0u, 0u,
				                   cap, (const SegCtl *)d_ctl, d_cur, d_ovf, ka, (const KT *)nullptr);
			else if (which == 1) {
				if constexpr (sizeof(KT) == 4)
					hipLaunchKernelGGL((rsx_pass32s_kernel<KT, 2, true, Pass32sCfgT<CFG::KPT, false>>), dim3(256), dim3(1024), 0, 0, (const KT *)d_in, (u64)n, d_slots, 0u, 0u,
					                   0u, cap, (const SegCtl *)d_ctl, d_cur, d_ovf, ka, (const KT *)d_spl);
			} else
				hipLaunchKernelGGL((rsx_pass32s_kernel<KT, 1, false, CFG>), dim3(256), dim3(1024), 0, 0, (const KT *)d_in, (u64)n, d_slots, 0u, 0u, 0u,
				                   cap, (const SegCtl *)d_ctl, d_cur, d_ovf, ka, (const KT *)nullptr);
			CK(hipEventRecord(e1, 0));
			CK(hipEventSynchronize(e1));
			float ms;
			CK(hipEventElapsedTime(&ms, e0, e1));
			CK(hipGetLastError());
			best = std::min(best, ms);
			CK(hipEventDestroy(e0));
			CK(hipEventDestroy(e1));
		}
		u32 ovf = 0;
		CK(hipMemcpy(&ovf, d_ovf, 4, hipMemcpyDeviceToHost));
		u64 got[4];
		CK(hipMemset(d_chk, 0, 64));
		hipLaunchKernelGGL((check_kernel<KT>), dim3(256), dim3(1024), 0, 0, (const KT *)d_slots, (const u32 *)d_cur, cap, PASS32S_BACK, (const KT *)d_spl,
		                   d_chk);
		CK(hipMemcpy(got, d_chk, 32, hipMemcpyDeviceToHost));
		std::vector<u32> cur(512);
		CK(hipMemcpy(cur.data(), d_cur, 2048, hipMemcpyDeviceToHost));
		u32 mx = 0;
		for (u32 d = 0; d < 256; ++d)
			mx = std::max(mx, cur[d] + cur[256 + d]);
		printf("%-46s %.3f ms = %.0f GB/s   overflow %u, misplaced %llu, keys %llu (%s), sum %s, mix %s; largest bucket %.2f x the mean\n", name, best,
		       n * 2.0 * sizeof(KT) / best / 1e6, ovf, (unsigned long long)got[0], (unsigned long long)got[3], got[3] == n ? "all" : "NOT ALL",
		       got[1] == want[1] ? "ok" : "DIFFERENT", got[2] == want[2] ? "ok" : "DIFFERENT", (double)mx / mean);
		fflush(stdout);
	};
	if (mode == 0) {
		run("by the top byte (rsx_pass32s_kernel)", 0);
		run("by the top byte, no prefetch", 2);
	}
	if (sizeof(KT) == 4)
		run("by 255 splitters searched in the LDS", 1);
	if (mode == 0)
		run("by the top byte (again)", 0);
	return 0;
}

int main(int argc, char **argv)
{
	const int log2n = argc > 1 ? atoi(argv[1]) : 28;
	const u32 mode = argc > 2 ? (u32)atoi(argv[2]) : 0;
	const int key_bytes = argc > 3 ? atoi(argv[3]) : 4;
	// (argv[4] = 0: a second returning atomic in the staging phase instead of the kept rank, Pass32sCfgT<.., false>)
	const bool rank1 = argc > 4 ? atoi(argv[4]) != 0 : true;
	if (key_bytes == 8)
		return rank1 ? run_all<u64, Pass32sCfgT<14>>(log2n, 0) : run_all<u64, Pass32sCfgT<14, false>>(log2n, 0);
	return rank1 ? run_all<u32, Pass32sCfg>(log2n, mode) : run_all<u32, Pass32sCfgT<28, false>>(log2n, mode);
}
