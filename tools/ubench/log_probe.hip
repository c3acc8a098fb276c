// log_probe: the kernels of csrc/rsx_logroute.hpp on their own -- 8-byte keys by (bit length, mantissa) digits: sample, histogram,
// plan, level-1 pass (exact buckets), level-2 pass (four-byte slots), the small keys written out, the leaves -- each timed with
// HIP events, the control block printed, the result compared with std::sort on the host (up to 2^26 keys) or checked for order
// and checksums on the device.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I radix_sorting_amd/csrc tools/ubench/log_probe.hip -o tools/ubench/log_probe.bin
// Run:   log_probe.bin [log2 n = 28] [bmax = 40] [reps = 5] [n offset = 0]      (LOG_PROBE_BIG=1: the leaves' 10240-value shape)
#include "rsx_logroute.hpp"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
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

// SURVEY.md 8d cfg 3 (iv): key = 2^(b-1) + (r & (2^(b-1) - 1)), b = 1 + (r >> 58) % bmax, r = splitmix64 of the index
__global__ void gen_kernel(u64 *k, u64 n, u32 bmax, u64 seed)
{
	for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) {
		u64 z = (seed + i + 1) * 0x9E3779B97F4A7C15ull;
		z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
		z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
		z ^= z >> 31;
		const u32 b = 1 + (u32)((z >> 58) % bmax);
		k[i] = ((u64)1 << (b - 1)) + (z & (((u64)1 << (b - 1)) - 1));
	}
}

__global__ void check_kernel(const u64 *a, u64 n, u64 *out)
{
	u64 desc = 0, sum = 0, mix = 0;
	for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) {
		const u64 k = a[i];
		if (i + 1 < n && a[i + 1] < k)
			++desc;
		sum += k;
		mix ^= (k + 0x9E3779B97F4A7C15ull) * 0xBF58476D1CE4E5B9ull;
	}
	atomicAdd((unsigned long long *)&out[0], desc);
	atomicAdd((unsigned long long *)&out[1], sum);
	atomicXor((unsigned long long *)&out[2], mix);
}

int main(int argc, char **argv)
{
	const int log2n = argc > 1 ? atoi(argv[1]) : 28;
	const u32 bmax = argc > 2 ? (u32)atoi(argv[2]) : 40;
	const int reps = argc > 3 ? atoi(argv[3]) : 5;
	const u64 n = ((u64)1 << log2n) + (argc > 4 ? (u64)atoll(argv[4]) : 0);
	typedef u64 KT;
	typedef LogP2Cfg P2;
	KT *src, *aux, *keep;
	CK(hipMalloc(&src, n * 8));
	CK(hipMalloc(&aux, n * 8));
	CK(hipMalloc(&keep, n * 8));
	const size_t tiles_cap = n / P2::TILE + 257;
	const size_t cur2_off = sizeof(LogCtl), tabs_off = cur2_off + 2 * 65536 * sizeof(u32);
	const size_t tiles_off = (tabs_off + sizeof(LogTabs) + 255) & ~(size_t)255;
	const size_t zero_bytes = tabs_off + offsetof(LogTabs, offs_small);
	const size_t l2_cap = n + n / 8 + (size_t)65536 * 700;
	char *logb;
	u32 *slots;
	CK(hipMalloc(&logb, tiles_off + tiles_cap * sizeof(LogTile)));
	CK(hipMalloc(&slots, (l2_cap + P2::TILE + 64) * sizeof(u32)));
	LogCtl *ctl = (LogCtl *)logb;
	u32 *cur2 = (u32 *)(logb + cur2_off);
	LogTabs *tabs = (LogTabs *)(logb + tabs_off);
	LogTile *tiles = (LogTile *)(logb + tiles_off);
	Plan *hplan;
	CK(hipHostMalloc((void **)&hplan, sizeof(Plan), hipHostMallocMapped));
	Plan *dplan;
	CK(hipHostGetDevicePointer((void **)&dplan, hplan, 0));
	u64 *chk;
	CK(hipMalloc(&chk, 6 * 8));
	const KdfArgs<KT> ka{0, 0, 0};
	const bool big = getenv("LOG_PROBE_BIG") != nullptr || log2n > 28;
	gen_kernel<<<2048, 256>>>(keep, n, bmax, 12345);
	CK(hipDeviceSynchronize());
	hipEvent_t ev[9];
	for (auto &e : ev)
		CK(hipEventCreate(&e));
	const char *names[8] = {"zero+sample", "hist", "plan", "pass1", "pass2", "fill", "leaves", "total"};
	double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
	LogCtl h;
	for (int rep = 0; rep < reps + 1; ++rep) {
		CK(hipMemcpy(src, keep, n * 8, hipMemcpyDeviceToDevice));
		CK(hipMemset(aux, 0x5A, n * 8));
		CK(hipDeviceSynchronize());
		CK(hipEventRecord(ev[0]));
		CK(hipMemsetAsync(logb, 0, zero_bytes));
		rsx_log_sample_kernel<KT><<<1, 1024>>>(src, n, ka, ctl, big ? LOG_LEAF_CAP_BIG : LOG_LEAF_CAP);
		CK(hipEventRecord(ev[1]));
		rsx_log_hist_kernel<KT><<<512, 1024>>>(src, n, ka, ctl, tabs);
		CK(hipEventRecord(ev[2]));
		rsx_log_plan_kernel<<<1, 1024>>>(ctl, tabs, tiles, n, (u32)n, (u32)l2_cap, (u32)tiles_cap, (u32)P2::TILE, (u32)P2::GRID, nullptr, dplan);
		CK(hipEventRecord(ev[3]));
		rsx_log_pass1_kernel<KT><<<256, LogP1Cfg::BLOCK>>>(src, n, aux, ctl, tabs, ka);
		CK(hipEventRecord(ev[4]));
		rsx_log_pass2_kernel<KT><<<P2::GRID, P2::BLOCK>>>(aux, slots, tiles, ctl, tabs, cur2, (u32)l2_cap, ka);
		CK(hipEventRecord(ev[5]));
		rsx_log_fill_kernel<KT><<<2048, LOG_FILL_BLOCK>>>(src, aux, ctl, tabs, ka);
		CK(hipEventRecord(ev[6]));
		if (big)
			rsx_log_leaf_kernel<KT, LogLeafCfgBig><<<65536, LogLeafCfgBig::BLOCK>>>(src, aux, slots, ctl, tabs, cur2, ka, 0u, 256u);
		else
			rsx_log_leaf_kernel<KT, LogLeafCfg><<<65536, LogLeafCfg::BLOCK>>>(src, aux, slots, ctl, tabs, cur2, ka, 0u, 256u);
		CK(hipEventRecord(ev[7]));
		CK(hipGetLastError());
		CK(hipDeviceSynchronize());
		CK(hipMemcpy(&h, ctl, sizeof h, hipMemcpyDeviceToHost));
		if (rep == 0)
			continue;   // warm-up
		for (int i = 0; i < 7; ++i) {
			float ms;
			CK(hipEventElapsedTime(&ms, ev[i], ev[i + 1]));
			acc[i] += ms;
		}
		float ms;
		CK(hipEventElapsedTime(&ms, ev[0], ev[7]));
		acc[7] += ms;
	}
	printf("n = %llu bmax = %u: go %u B %u m %u ndig %u fail %u desc %u or %08x%08x nsmall %u ntiles2 %u ok %u sorted %u per2 %u maxh1 %u ncols %u\n",
	       (unsigned long long)n, bmax, h.go, h.B, h.m, h.ndig, h.fail, h.desc_cnt, h.or_hi, h.or_lo, h.nsmall, h.ntiles2, h.ok, h.sorted,
	       h.per2, h.maxh1, h.ncols);
	for (int i = 0; i < 8; ++i)
		printf("  %-12s %8.3f ms\n", names[i], acc[i] / reps);
	const double nbig = (double)n - h.nsmall;
	printf("  GB/s: hist %.0f  pass1 %.0f  pass2 %.0f  fill %.0f  leaves %.0f\n", n * 8.0 / (acc[1] / reps) / 1e6,
	       (n * 8.0 + nbig * 8.0) / (acc[3] / reps) / 1e6, nbig * 12.0 / (acc[4] / reps) / 1e6, h.nsmall * 8.0 / (acc[5] / reps) / 1e6,
	       nbig * 12.0 / (acc[6] / reps) / 1e6);
	if (!h.ok || h.fail) {
		printf("the route did not run\n");
		return 1;
	}
	if (getenv("LOG_PROBE_LEAF_RANGES")) {
		// the leaves of four exponents at a time (a level-1 digit is (bit length - 13) << m | mantissa bits): where their time goes
		for (u32 e0 = 0; (e0 << h.m) < h.ndig; e0 += 4) {
			CK(hipEventRecord(ev[0]));
			rsx_log_leaf_kernel<KT, LogLeafCfg><<<65536, LogLeafCfg::BLOCK>>>(src, aux, slots, ctl, tabs, cur2, ka, e0 << h.m, (e0 + 4) << h.m);
			CK(hipEventRecord(ev[1]));
			CK(hipDeviceSynchronize());
			float ms;
			CK(hipEventElapsedTime(&ms, ev[0], ev[1]));
			printf("  leaves of bit lengths %2u..%2u: %.3f ms\n", LOG_C + 1 + e0, LOG_C + 4 + e0, ms);
		}
	}
	const KT *res = (h.ncols & 1) ? aux : src;
	CK(hipMemset(chk, 0, 6 * 8));
	check_kernel<<<2048, 256>>>(keep, n, chk);
	check_kernel<<<2048, 256>>>(res, n, chk + 3);
	u64 c[6];
	CK(hipMemcpy(c, chk, sizeof c, hipMemcpyDeviceToHost));
	printf("device check: descents %llu, sum %s, mix %s\n", (unsigned long long)c[3], c[1] == c[4] ? "kept" : "CHANGED", c[2] == c[5] ? "kept" : "CHANGED");
	int rc = (c[3] || c[1] != c[4] || c[2] != c[5]) ? 1 : 0;
	if (log2n <= 26) {
		std::vector<u64> a(n), b(n);
		CK(hipMemcpy(a.data(), keep, n * 8, hipMemcpyDeviceToHost));
		CK(hipMemcpy(b.data(), res, n * 8, hipMemcpyDeviceToHost));
		std::sort(a.begin(), a.end());
		u64 bad = 0, first = ~0ull;
		for (u64 i = 0; i < n; ++i)
			if (a[i] != b[i]) {
				if (!bad)
					first = i;
				++bad;
			}
		printf("host check: %llu of %llu places differ from std::sort%s\n", (unsigned long long)bad, (unsigned long long)n, bad ? "" : " (none)");
		if (bad) {
			printf("  first at %llu: want %llx got %llx\n", (unsigned long long)first, (unsigned long long)a[first], (unsigned long long)b[first]);
			rc = 1;
		}
	}
	return rc;
}
