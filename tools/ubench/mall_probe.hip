// mall_probe: does the 256 MB Infinity Cache keep what a kernel has just WRITTEN, so that the next kernel reads it from there?
// (The question behind running the level-2 pass and the leaves group by group: the level-2 slots are 0.5 GiB written and 0.5 GiB
// read back.)  For a range of sizes: kernel W writes the buffer, kernel R reads it (sum), both timed with HIP events; "cold" =
// a 1 GiB buffer is written between W and R.  Read bandwidth well above the HBM's ~5 TB/s means the data came from the cache.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench/mall_probe.hip -o tools/ubench/mall_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned int u32;
typedef unsigned long long u64;
typedef u32 u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ __launch_bounds__(1024) void write_kernel(u32x4 *p, u64 nvec, u32 v)
{
	for (u64 i = (u64)blockIdx.x * 1024 + threadIdx.x; i < nvec; i += (u64)gridDim.x * 1024)
		p[i] = u32x4{v, v + 1, v + 2, (u32)i};
}
__global__ __launch_bounds__(1024) void read_kernel(const u32x4 *p, u64 nvec, u32 *out)
{
	u32 s = 0;
	for (u64 i = (u64)blockIdx.x * 1024 + threadIdx.x; i < nvec; i += (u64)gridDim.x * 1024) {
		const u32x4 x = p[i];
		s += x[0] ^ x[1] ^ x[2] ^ x[3];
	}
	if (s == 0x12345678u)
		*out = s;
}
int main()
{
	const u64 GiB = 1ull << 30;
	u32x4 *buf, *other;
	u32 *out;
	CK(hipMalloc(&buf, GiB));
	CK(hipMalloc(&other, GiB));
	CK(hipMalloc(&out, 4));
	hipEvent_t e[4];
	for (auto &x : e)
		CK(hipEventCreate(&x));
	printf("%8s | %10s %10s | %10s %10s | %10s\n", "MiB", "write ms", "GB/s", "read ms", "GB/s", "cold read GB/s");
	for (u64 mib : {32, 64, 96, 128, 192, 256, 384, 512, 1024}) {
		const u64 bytes = mib << 20, nvec = bytes / 16;
		float tw = 0, tr = 0, tc = 0;
		const int reps = 5;
		for (int r = 0; r < reps + 1; ++r) {
			write_kernel<<<1024, 1024>>>(other, GiB / 16, r);   // (the cache full of something else)
			CK(hipEventRecord(e[0]));
			write_kernel<<<1024, 1024>>>(buf, nvec, r);
			CK(hipEventRecord(e[1]));
			read_kernel<<<1024, 1024>>>(buf, nvec, out);
			CK(hipEventRecord(e[2]));
			write_kernel<<<1024, 1024>>>(other, GiB / 16, r);
			CK(hipEventRecord(e[3]));
			CK(hipDeviceSynchronize());
			float a, b;
			CK(hipEventElapsedTime(&a, e[0], e[1]));
			CK(hipEventElapsedTime(&b, e[1], e[2]));
			if (r) { tw += a; tr += b; }
			CK(hipEventRecord(e[0]));
			read_kernel<<<1024, 1024>>>(buf, nvec, out);
			CK(hipEventRecord(e[1]));
			CK(hipDeviceSynchronize());
			CK(hipEventElapsedTime(&a, e[0], e[1]));
			if (r) tc += a;
		}
		tw /= reps; tr /= reps; tc /= reps;
		printf("%8llu | %10.4f %10.0f | %10.4f %10.0f | %10.0f\n", (unsigned long long)mib, tw, bytes / tw / 1e6, tr, bytes / tr / 1e6, bytes / tc / 1e6);
	}
	return 0;
}
