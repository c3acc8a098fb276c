// copy_cold: what a plain copy kernel reaches when it never sees a warm cache -- eight 1 GiB buffers, every launch copies the next
// pair (source and destination both untouched for seven launches: nothing of them in the 256 MB Infinity Cache, which holds the
// previous launch's dirty lines), against the same kernel copying ONE pair again and again.  The state in which every pass of a
// sort finds the memory system.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench/copy_cold.hip -o tools/ubench/copy_cold.bin
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>

#define CK(x)                                                                             \
	do {                                                                                  \
		hipError_t e_ = (x);                                                              \
		if (e_ != hipSuccess) {                                                           \
			printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
			exit(1);                                                                      \
		}                                                                                 \
	} while (0)

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int U> __global__ __launch_bounds__(1024) void copy_kernel(const u32x4 *__restrict__ src, u32x4 *__restrict__ dst, size_t nvec)
{
	const size_t stride = (size_t)gridDim.x * 1024 * U;
	for (size_t v0 = (size_t)blockIdx.x * 1024 * U + threadIdx.x; v0 < nvec; v0 += stride) {
		u32x4 x[U];
#pragma unroll
		for (int u = 0; u < U; ++u)
			if (v0 + (size_t)u * 1024 < nvec)
				x[u] = src[v0 + (size_t)u * 1024];
#pragma unroll
		for (int u = 0; u < U; ++u)
			if (v0 + (size_t)u * 1024 < nvec)
				dst[v0 + (size_t)u * 1024] = x[u];
	}
}
template <int U> __global__ __launch_bounds__(1024) void read_kernel(const u32x4 *__restrict__ src, u32x4 *__restrict__ dst, size_t nvec)
{
	const size_t stride = (size_t)gridDim.x * 1024 * U;
	u32x4 acc = {0, 0, 0, 0};
	for (size_t v0 = (size_t)blockIdx.x * 1024 * U + threadIdx.x; v0 < nvec; v0 += stride) {
#pragma unroll
		for (int u = 0; u < U; ++u)
			if (v0 + (size_t)u * 1024 < nvec)
				acc ^= src[v0 + (size_t)u * 1024];
	}
	if (acc[0] == 0x12345u && acc[1] == 7u)
		dst[threadIdx.x] = acc;
}
template <int U> __global__ __launch_bounds__(1024) void write_kernel(const u32x4 *__restrict__ src, u32x4 *__restrict__ dst, size_t nvec)
{
	const size_t stride = (size_t)gridDim.x * 1024 * U;
	const u32x4 x = {(unsigned)blockIdx.x, 1, 2, 3};
	for (size_t v0 = (size_t)blockIdx.x * 1024 * U + threadIdx.x; v0 < nvec; v0 += stride) {
#pragma unroll
		for (int u = 0; u < U; ++u)
			if (v0 + (size_t)u * 1024 < nvec)
				dst[v0 + (size_t)u * 1024] = x;
	}
}

// the same write with NON-TEMPORAL stores (the compiler's nt bit: streaming, not kept in the caches)
template <int U> __global__ __launch_bounds__(1024) void write_nt_kernel(const u32x4 *__restrict__ src, u32x4 *__restrict__ dst, size_t nvec)
{
	const size_t stride = (size_t)gridDim.x * 1024 * U;
	const u32x4 x = {(unsigned)blockIdx.x, 1, 2, 3};
	for (size_t v0 = (size_t)blockIdx.x * 1024 * U + threadIdx.x; v0 < nvec; v0 += stride) {
#pragma unroll
		for (int u = 0; u < U; ++u)
			if (v0 + (size_t)u * 1024 < nvec)
				__builtin_nontemporal_store(x, &dst[v0 + (size_t)u * 1024]);
	}
}
template <int U> __global__ __launch_bounds__(1024) void copy_nt_kernel(const u32x4 *__restrict__ src, u32x4 *__restrict__ dst, size_t nvec)
{
	const size_t stride = (size_t)gridDim.x * 1024 * U;
	for (size_t v0 = (size_t)blockIdx.x * 1024 * U + threadIdx.x; v0 < nvec; v0 += stride) {
		u32x4 x[U];
#pragma unroll
		for (int u = 0; u < U; ++u)
			if (v0 + (size_t)u * 1024 < nvec)
				x[u] = __builtin_nontemporal_load(&src[v0 + (size_t)u * 1024]);
#pragma unroll
		for (int u = 0; u < U; ++u)
			if (v0 + (size_t)u * 1024 < nvec)
				__builtin_nontemporal_store(x[u], &dst[v0 + (size_t)u * 1024]);
	}
}

// the buffer read from its end to its beginning (workgroup b takes the b-th stretch from the END)
template <int U> __global__ __launch_bounds__(1024) void read_backward_kernel(const u32x4 *__restrict__ src, u32x4 *__restrict__ dst, size_t nvec)
{
	const size_t stride = (size_t)gridDim.x * 1024 * U;
	u32x4 acc = {0, 0, 0, 0};
	for (size_t v0 = (size_t)blockIdx.x * 1024 * U + threadIdx.x; v0 < nvec; v0 += stride) {
#pragma unroll
		for (int u = 0; u < U; ++u)
			if (v0 + (size_t)u * 1024 < nvec)
				acc ^= src[nvec - 1 - (v0 + (size_t)u * 1024)];
	}
	if (acc[0] == 0x12345u && acc[1] == 7u)
		dst[threadIdx.x] = acc;
}

int main()
{
	const size_t bytes = (size_t)1 << 30, nvec = bytes / 16;
	const int NB = 9;
	u32x4 *buf[NB];
	for (int i = 0; i < NB; ++i) {
		CK(hipMalloc(&buf[i], bytes));
		CK(hipMemset(buf[i], i + 1, bytes));
	}
	auto run = [&](const char *name, auto kernel, double bytes_moved, bool rotate, unsigned grid) {
		float best = 1e9f, sum = 0;
		const int reps = 16;
		for (int r = 0; r < reps + 2; ++r) {
			const int a = rotate ? (2 * r) % NB : 0, b = rotate ? (2 * r + 1) % NB : 1;
			hipEvent_t e0, e1;
			CK(hipEventCreate(&e0));
			CK(hipEventCreate(&e1));
			CK(hipEventRecord(e0, 0));
			hipLaunchKernelGGL(kernel, dim3(grid), dim3(1024), 0, 0, (const u32x4 *)buf[a], buf[b], nvec);
			CK(hipEventRecord(e1, 0));
			CK(hipEventSynchronize(e1));
			float ms;
			CK(hipEventElapsedTime(&ms, e0, e1));
			if (r >= 2) {
				best = std::min(best, ms);
				sum += ms;
			}
			CK(hipEventDestroy(e0));
			CK(hipEventDestroy(e1));
		}
		printf("%-34s %-28s grid %4u: best %.3f ms = %.0f GB/s, mean %.3f ms = %.0f GB/s\n", name,
		       rotate ? "eight buffers in turn (cold)" : "one pair again and again", grid, best, bytes_moved / best / 1e6, sum / reps,
		       bytes_moved / (sum / reps) / 1e6);
	};
	// ... and behind a kernel that has just WRITTEN a gigabyte elsewhere (what precedes every kernel of a sort): the Infinity Cache
	// is full of dirty lines, which the timed kernel's misses evict
	auto run_behind_write = [&](const char *name, auto kernel, double bytes_moved, unsigned grid) {
		float best = 1e9f, sum = 0;
		const int reps = 16;
		for (int r = 0; r < reps + 2; ++r) {
			const int w = (3 * r) % NB, a = (3 * r + 1) % NB, b = (3 * r + 2) % NB;
			hipLaunchKernelGGL(write_kernel<4>, dim3(2048), dim3(1024), 0, 0, (const u32x4 *)buf[a], buf[w], nvec);
			hipEvent_t e0, e1;
			CK(hipEventCreate(&e0));
			CK(hipEventCreate(&e1));
			CK(hipEventRecord(e0, 0));
			hipLaunchKernelGGL(kernel, dim3(grid), dim3(1024), 0, 0, (const u32x4 *)buf[a], buf[b], nvec);
			CK(hipEventRecord(e1, 0));
			CK(hipEventSynchronize(e1));
			float ms;
			CK(hipEventElapsedTime(&ms, e0, e1));
			if (r >= 2) {
				best = std::min(best, ms);
				sum += ms;
			}
			CK(hipEventDestroy(e0));
			CK(hipEventDestroy(e1));
		}
		printf("%-34s %-28s grid %4u: best %.3f ms = %.0f GB/s, mean %.3f ms = %.0f GB/s\n", name, "behind a 1 GiB write elsewhere", grid, best,
		       bytes_moved / best / 1e6, sum / reps, bytes_moved / (sum / reps) / 1e6);
	};
	// ... and a read of the very buffer the kernel before has written, from its beginning and from its end: does the cache still hold
	// what was written last?
	auto run_own = [&](const char *name, auto kernel, unsigned grid) {
		float best = 1e9f, sum = 0;
		const int reps = 16;
		for (int r = 0; r < reps + 2; ++r) {
			const int w = (2 * r) % NB, b = (2 * r + 1) % NB;
			hipLaunchKernelGGL(write_kernel<4>, dim3(512), dim3(1024), 0, 0, (const u32x4 *)buf[b], buf[w], nvec);
			hipEvent_t e0, e1;
			CK(hipEventCreate(&e0));
			CK(hipEventCreate(&e1));
			CK(hipEventRecord(e0, 0));
			hipLaunchKernelGGL(kernel, dim3(grid), dim3(1024), 0, 0, (const u32x4 *)buf[w], buf[b], nvec);
			CK(hipEventRecord(e1, 0));
			CK(hipEventSynchronize(e1));
			float ms;
			CK(hipEventElapsedTime(&ms, e0, e1));
			if (r >= 2) {
				best = std::min(best, ms);
				sum += ms;
			}
			CK(hipEventDestroy(e0));
			CK(hipEventDestroy(e1));
		}
		printf("%-34s %-28s grid %4u: best %.3f ms = %.0f GB/s, mean %.3f ms = %.0f GB/s\n", name, "the buffer just written", grid, best,
		       1.0 * bytes / best / 1e6, sum / reps, 1.0 * bytes / (sum / reps) / 1e6);
	};
	// ... behind a writer whose stores are non-temporal
	auto run_behind = [&](const char *name, const char *what, auto pre, auto kernel, double bytes_moved, unsigned grid) {
		float best = 1e9f, sum = 0;
		const int reps = 16;
		for (int r = 0; r < reps + 2; ++r) {
			const int w = (3 * r) % NB, a = (3 * r + 1) % NB, b = (3 * r + 2) % NB;
			hipLaunchKernelGGL(pre, dim3(2048), dim3(1024), 0, 0, (const u32x4 *)buf[a], buf[w], nvec);
			hipEvent_t e0, e1;
			CK(hipEventCreate(&e0));
			CK(hipEventCreate(&e1));
			CK(hipEventRecord(e0, 0));
			hipLaunchKernelGGL(kernel, dim3(grid), dim3(1024), 0, 0, (const u32x4 *)buf[a], buf[b], nvec);
			CK(hipEventRecord(e1, 0));
			CK(hipEventSynchronize(e1));
			float ms;
			CK(hipEventElapsedTime(&ms, e0, e1));
			if (r >= 2) {
				best = std::min(best, ms);
				sum += ms;
			}
			CK(hipEventDestroy(e0));
			CK(hipEventDestroy(e1));
		}
		printf("%-34s %-28s grid %4u: best %.3f ms = %.0f GB/s, mean %.3f ms = %.0f GB/s\n", name, what, grid, best, bytes_moved / best / 1e6,
		       sum / reps, bytes_moved / (sum / reps) / 1e6);
	};
	run_behind("read 1 GiB", "behind a NON-TEMPORAL 1 GiB write", write_nt_kernel<4>, read_kernel<4>, 1.0 * bytes, 512);
	run_behind("copy 1 GiB -> 1 GiB (2 GiB moved)", "behind a NON-TEMPORAL 1 GiB write", write_nt_kernel<4>, copy_kernel<4>, 2.0 * bytes, 512);
	run_behind("copy, nt loads and stores", "behind a NON-TEMPORAL copy", copy_nt_kernel<4>, copy_nt_kernel<4>, 2.0 * bytes, 512);
	run_behind("copy 1 GiB -> 1 GiB (2 GiB moved)", "behind an ordinary copy", copy_kernel<4>, copy_kernel<4>, 2.0 * bytes, 512);
	run("write 1 GiB, non-temporal", write_nt_kernel<4>, 1.0 * bytes, true, 512);
	run("copy, nt loads and stores", copy_nt_kernel<4>, 2.0 * bytes, true, 512);
	run_own("read 1 GiB, forward", read_kernel<4>, 512);
	run_own("read 1 GiB, from its end", read_backward_kernel<4>, 512);
	for (unsigned grid : {512u, 2048u}) {
		run_behind_write("copy 1 GiB -> 1 GiB (2 GiB moved)", copy_kernel<4>, 2.0 * bytes, grid);
		run_behind_write("read 1 GiB", read_kernel<4>, 1.0 * bytes, grid);
		run_behind_write("write 1 GiB", write_kernel<4>, 1.0 * bytes, grid);
	}
	for (unsigned grid : {512u, 2048u}) {
		run("copy 1 GiB -> 1 GiB (2 GiB moved)", copy_kernel<4>, 2.0 * bytes, false, grid);
		run("copy 1 GiB -> 1 GiB (2 GiB moved)", copy_kernel<4>, 2.0 * bytes, true, grid);
		run("read 1 GiB", read_kernel<4>, 1.0 * bytes, false, grid);
		run("read 1 GiB", read_kernel<4>, 1.0 * bytes, true, grid);
		run("write 1 GiB", write_kernel<4>, 1.0 * bytes, false, grid);
		run("write 1 GiB", write_kernel<4>, 1.0 * bytes, true, grid);
	}
	return 0;
}
