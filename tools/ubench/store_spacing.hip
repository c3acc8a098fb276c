// store_spacing: is the ceiling of a pass's scattered stores address TRANSLATION or the memory side?  (round-4 review, item 9)
// 1 GiB is written as 2 M runs of 512 bytes, 256 runs per "tile" (one per digit), 16-byte stores, tiles handed out by a ticket
// -- the store pattern of rsx_scatter2_kernel's level-1 pass on 2^28 u32 keys -- with the 256 destinations of a tile
//   (i)   4 MiB apart, as in the sort (stream r = bytes [r * 4 MiB, (r + 1) * 4 MiB), tile t at offset t * 512),
//   (ii)  inside ONE 2 MiB window: window t / 16 holds the 256 streams' pieces of 16 consecutive tiles, 8 KiB each,
//   (iii) 64 MiB apart (a 16 GiB allocation of which every stream uses its first 4 MiB),
// each with run starts that are 4-byte-, 64-byte- and 128-byte-aligned (misalign = 4 x (r & 3) + 4, 0 mod 64, 0 mod 128).
// If (ii) is markedly faster than (i), translation (256 pages in flight per workgroup) is the ceiling and slots interleaved at
// 2 MiB granularity are the lever; if not, it is the memory side's appetite for short runs.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench/store_spacing.hip -o tools/ubench/store_spacing.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <algorithm>

typedef unsigned int u32;
typedef unsigned long long u64;
typedef u32 u32x4 __attribute__((ext_vector_type(4)));
typedef u32x4 uu32x4 __attribute__((aligned(4)));

#define CK(x)                                                                             \
	do {                                                                                  \
		hipError_t e_ = (x);                                                              \
		if (e_ != hipSuccess) {                                                           \
			printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
			exit(1);                                                                      \
		}                                                                                 \
	} while (0)

// layout 0: stream-major (stride_elems between streams); layout 1: 2 MiB windows of 16 tiles
__global__ __launch_bounds__(1024) void k(u32 *out, u64 stride_elems, u32 ntiles, u32 layout, u32 mis_mode, u32 *ticket)
{
	__shared__ u32 s_t;
	const u32 tid = threadIdx.x;
	constexpr u32 RUN = 128;   // elements: 512 bytes
	for (;;) {
		if (tid == 0)
			s_t = atomicAdd(ticket, 1u);
		__syncthreads();
		const u32 t = s_t;
		__syncthreads();
		if (t >= ntiles)
			return;
		for (u32 i0 = tid * 4; i0 < 256 * RUN; i0 += 1024 * 4) {
			const u32 r = i0 / RUN, o = i0 % RUN;
			// where run (t, r) starts, in elements; the misalignment shifts the whole stream, so consecutive tiles' runs of a
			// stream still abut (as the runs of one digit do in the sort)
			const u32 mis = mis_mode == 0 ? 1 + (r & 3) : 0;   // 4 .. 16 bytes off / aligned
			u64 e;
			if (layout == 0)
				e = (u64)r * stride_elems + (u64)t * RUN;
			else
				e = (u64)(t / 16) * (256 * 16 * RUN) + (u64)r * (16 * RUN) + (u64)(t % 16) * RUN;
			if (mis_mode == 1)
				e += 0;   // 512-byte aligned runs (a fortiori 64 and 128)
			u32 *dst = out + e + mis + o;
			*(uu32x4 *)dst = u32x4{i0, t, r, o};
		}
	}
}

// runs whose length varies (as in the sort: 128 +- 11 keys), so that runs start anywhere: the prefix within the stream is a hash
__global__ __launch_bounds__(1024) void kv(u32 *out, u64 stride_elems, u32 ntiles, u32 layout, u32 align_elems, u32 *ticket)
{
	__shared__ u32 s_t;
	const u32 tid = threadIdx.x;
	constexpr u32 RUN = 128;
	for (;;) {
		if (tid == 0)
			s_t = atomicAdd(ticket, 1u);
		__syncthreads();
		const u32 t = s_t;
		__syncthreads();
		if (t >= ntiles)
			return;
		for (u32 i0 = tid * 4; i0 < 256 * RUN; i0 += 1024 * 4) {
			const u32 r = i0 / RUN, o = i0 % RUN;
			// a pseudo-random start offset of 0 .. 15 elements (x align) inside a 512 + 64-byte pitch: runs do not abut exactly,
			// every run has two ragged ends of its own unless align_elems makes them whole atoms
			u32 h = (t * 2654435761u) ^ (r * 40503u);
			h ^= h >> 15;
			const u32 off = align_elems >= 16 ? 0u : (h & 15u) / align_elems * align_elems;
			u64 e;
			if (layout == 0)
				e = (u64)r * stride_elems + (u64)t * (RUN + 16);
			else
				e = (u64)(t / 16) * (256 * 16 * (RUN + 16)) + (u64)r * (16 * (RUN + 16)) + (u64)(t % 16) * (RUN + 16);
			u32 *dst = out + e + off + o;
			*(uu32x4 *)dst = u32x4{i0, t, r, o};
		}
	}
}

int main()
{
	const u64 total_elems = 1ull << 28;   // 1 GiB
	const u32 ntiles = (u32)(total_elems / (256ull * 128));
	u32 *d, *d_ticket;
	const u64 big = 256ull * (80ull << 20);   // layout (iii) with ragged runs: 256 streams 72 MiB apart
	CK(hipMalloc(&d, big));
	CK(hipMalloc(&d_ticket, 1024));
	CK(hipMemset(d, 0, big));   // (pages touched once: no first-touch cost inside the timings)
	struct Cfg {
		const char *name;
		u64 stride;
		u32 layout;
	} cfgs[] = {{"(i)   256 streams 4 MiB apart (the sort)", total_elems / 256, 0},
	            {"(ii)  256 streams inside one 2 MiB window", 0, 1},
	            {"(iii) 256 streams 64 MiB apart", (64ull << 20) / 4, 0},
	            {"(i')  256 streams 4.5 MiB apart (+ slack, as the slots)", total_elems / 256 + total_elems / 2048, 0}};
	for (int pass = 0; pass < 2; ++pass) {
		printf(pass == 0 ? "-- abutting 512-byte runs (stream shifted by 4 .. 16 bytes / aligned)\n"
		                 : "-- runs with ragged ends of their own (pitch 576 bytes, start offset a multiple of 4 / 16 / 64 bytes)\n");
		for (auto &c : cfgs) {
			for (u32 m = 0; m < (pass == 0 ? 2u : 3u); ++m) {
				float best = 1e9f;
				for (int rep = 0; rep < 4; ++rep) {
					CK(hipMemset(d_ticket, 0, 1024));
					hipEvent_t e0, e1;
					CK(hipEventCreate(&e0));
					CK(hipEventCreate(&e1));
					CK(hipEventRecord(e0));
					if (pass == 0)
						hipLaunchKernelGGL(k, dim3(256), dim3(1024), 0, 0, d, c.stride, ntiles, c.layout, m, d_ticket);
					else
						hipLaunchKernelGGL(kv, dim3(256), dim3(1024), 0, 0, d, c.stride + c.stride / 8, ntiles, c.layout,
						                   m == 0 ? 1u : m == 1 ? 4u : 16u, d_ticket);
					CK(hipEventRecord(e1));
					CK(hipEventSynchronize(e1));
					float ms;
					CK(hipEventElapsedTime(&ms, e0, e1));
					CK(hipGetLastError());
					best = std::min(best, ms);
					CK(hipEventDestroy(e0));
					CK(hipEventDestroy(e1));
				}
				const char *what = pass == 0 ? (m == 0 ? "4-byte aligned " : "512-byte aligned") : (m == 0 ? "starts mod 4 B " : m == 1 ? "starts mod 16 B" : "starts mod 64 B");
				printf("%-58s %s: %.3f ms = %.0f GB/s\n", c.name, what, best, total_elems * 4.0 / best / 1e6);
			}
		}
	}
	return 0;
}
