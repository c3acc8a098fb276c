// gather_runs: how fast does the chip READ an array as scattered runs?  (store_runs.hip asked the same about writes.)
// A wave reads `run` bytes at a time from pseudo-randomly permuted places of a 1 GiB array (every run once), the places
// aligned to `align` bytes + `skew` bytes; the data is summed so that nothing is optimised away.  Compare: a linear read.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench/gather_runs.hip -o tools/ubench/gather_runs.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned int u32;
typedef unsigned long long u64;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// run_bytes: 128 .. 4096 (a power of two).  Lanes read 4 bytes each (element loads, as a gather of ragged pieces would).
__global__ __launch_bounds__(1024) void gather_kernel(const u32 *__restrict__ a, u64 nruns, u32 run_words, u32 skew_words, u64 *out, u32 mult)
{
	const u32 lane = threadIdx.x & 63;
	const u64 wave = ((u64)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((u64)gridDim.x * blockDim.x) >> 6;
	u32 acc = 0;
	for (u64 r0 = wave * 8; r0 < nruns; r0 += nwaves * 8) {
		// eight runs per wave and step, all their loads in flight
		u32 v[8][4];
#pragma unroll
		for (int k = 0; k < 8; ++k) {
			const u64 r = r0 + k;
			const u64 p = r < nruns ? (r * mult) % nruns : 0;   // mult odd, nruns a power of two: a permutation
			const u32 *q = a + p * run_words + skew_words;
#pragma unroll
			for (int j = 0; j < 4; ++j)
				v[k][j] = (lane + 64 * j < run_words && r < nruns) ? q[lane + 64 * j] : 0u;
		}
#pragma unroll
		for (int k = 0; k < 8; ++k)
#pragma unroll
			for (int j = 0; j < 4; ++j)
				acc += v[k][j];
	}
	if (acc == 0x12345678u)
		out[0] = acc;
}

int main()
{
	const u64 n = (u64)1 << 28;   // words: 1 GiB
	u32 *d;
	u64 *o;
	CK(hipMalloc(&d, n * 4 + 4096));
	CK(hipMalloc(&o, 8));
	CK(hipMemset(d, 1, n * 4 + 4096));
	for (u32 run_bytes : {128u, 256u, 512u, 1024u}) {
		for (u32 skew : {0u, 1u, 5u}) {
			const u32 run_words = run_bytes / 4;
			const u64 nruns = n / run_words;
			for (u32 mult : {1u, 0x9E3779B1u}) {
				float best = 1e9;
				for (int rep = 0; rep < 4; ++rep) {
					hipEvent_t e0, e1;
					CK(hipEventCreate(&e0));
					CK(hipEventCreate(&e1));
					CK(hipEventRecord(e0, 0));
					hipLaunchKernelGGL(gather_kernel, dim3(512), dim3(1024), 0, 0, (const u32 *)d, nruns, run_words, skew, o, mult);
					CK(hipEventRecord(e1, 0));
					CK(hipEventSynchronize(e1));
					float ms;
					CK(hipEventElapsedTime(&ms, e0, e1));
					best = ms < best ? ms : best;
				}
				printf("runs of %4u bytes, %s, start + %u words: %.3f ms = %.0f GB/s\n", run_bytes, mult == 1 ? "in order " : "scattered", skew, best,
				       n * 4.0 / best / 1e6);
			}
		}
	}
	return 0;
}
