// Store-path micro-benchmark: every workgroup writes "tiles" made of 256 runs of RUN bytes, run r of
// tile t going to base_r + t*RUN (256 output streams, like one radix pass), with 16-byte stores.
// Reports achieved GB/s vs run length and vs an unaligned start of each stream.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <algorithm>

typedef unsigned int u32;
typedef unsigned long long u64;
typedef u32 u32x4 __attribute__((ext_vector_type(4)));
typedef u32x4 uu32x4 __attribute__((aligned(4)));

// grid-stride over tiles; block = 512 threads; tile = 256 runs x run_bytes
// super-tile variant: a ticket covers `tps` consecutive tiles, written one after the other by the same workgroup
__global__ __launch_bounds__(512) void ks(u32 *out, u64 stream_stride_elems, u32 run_elems, u32 ntiles, u32 misalign, u32 *ticket,
                                          u32 tps, u32 delay)
{
	__shared__ u32 s_t;
	const u32 tid = threadIdx.x;
	for (;;) {
		if (tid == 0)
			s_t = atomicAdd(ticket, 1u);
		__syncthreads();
		const u32 st = s_t;
		__syncthreads();
		if (st * tps >= ntiles)
			return;
		const u32 tile_elems = 256 * run_elems;
		for (u32 t = st * tps; t < st * tps + tps && t < ntiles; ++t) {
			for (u32 i0 = tid * 4; i0 < tile_elems; i0 += 512 * 4) {
				const u32 r = i0 / run_elems, o = i0 % run_elems;
				u32 *dst = out + (u64)r * stream_stride_elems + misalign * (r & 3) + (u64)t * run_elems + o;
				*(uu32x4 *)dst = u32x4{i0, t, r, o};
			}
			for (u32 d = 0; d < delay; ++d)
				__builtin_amdgcn_s_sleep(64);
			__syncthreads();
		}
	}
}

__global__ __launch_bounds__(512) void k(u32 *out, u64 stream_stride_elems, u32 run_elems, u32 ntiles, u32 misalign, u32 *ticket)
{
	__shared__ u32 s_t;
	const u32 tid = threadIdx.x;
	for (;;) {
		if (tid == 0)
			s_t = atomicAdd(ticket, 1u);
		__syncthreads();
		const u32 t = s_t;
		__syncthreads();
		if (t >= ntiles)
			return;
		const u32 tile_elems = 256 * run_elems;
		for (u32 i0 = tid * 4; i0 < tile_elems; i0 += 512 * 4) {
			const u32 r = i0 / run_elems, o = i0 % run_elems;   // run, offset in run (run_elems multiple of 4)
			u32 *dst = out + (u64)r * stream_stride_elems + misalign * (r & 3) + (u64)t * run_elems + o;
			*(uu32x4 *)dst = u32x4{i0, t, r, o};
		}
	}
}

int main()
{
	const u64 total_elems = 1ull << 28;
	u32 *d, *d_ticket;
	hipMalloc(&d, total_elems * 4 + (1 << 20));
	hipMalloc(&d_ticket, 4);
	for (u32 misalign : {0u, 1u}) {
		for (u32 run_bytes : {32u, 64u, 128u, 256u, 512u, 1024u, 4096u}) {
			const u32 run_elems = run_bytes / 4;
			const u32 ntiles = (u32)(total_elems / (256ull * run_elems));
			const u64 stride = total_elems / 256;
			float best = 1e9;
			for (int rep = 0; rep < 3; ++rep) {
				hipMemset(d_ticket, 0, 4);
				hipEvent_t e0, e1;
				hipEventCreate(&e0);
				hipEventCreate(&e1);
				hipEventRecord(e0);
				hipLaunchKernelGGL(k, dim3(512), dim3(512), 0, 0, d, stride, run_elems, ntiles, misalign, d_ticket);
				hipEventRecord(e1);
				hipEventSynchronize(e1);
				float ms;
				hipEventElapsedTime(&ms, e0, e1);
				best = std::min(best, ms);
			}
			printf("misalign %u run %5u B: %.3f ms  %.0f GB/s\n", misalign, run_bytes, best, total_elems * 4.0 / (best * 1e-3) / 1e9);
		}
	}
	printf("-- super-tiles: same workgroup writes tps consecutive visits of each stream (misaligned)\n");
	for (u32 delay : {0u, 20u, 100u}) {
		for (u32 run_bytes : {128u, 256u}) {
			for (u32 tps : {1u, 2u, 4u, 8u, 16u}) {
				const u32 run_elems = run_bytes / 4;
				const u32 ntiles = (u32)(total_elems / (256ull * run_elems));
				const u64 stride = total_elems / 256;
				float best = 1e9;
				for (int rep = 0; rep < 3; ++rep) {
					hipMemset(d_ticket, 0, 4);
					hipEvent_t e0, e1;
					hipEventCreate(&e0);
					hipEventCreate(&e1);
					hipEventRecord(e0);
					hipLaunchKernelGGL(ks, dim3(512), dim3(512), 0, 0, d, stride, run_elems, ntiles, 1u, d_ticket, tps, delay);
					hipEventRecord(e1);
					hipEventSynchronize(e1);
					float ms;
					hipEventElapsedTime(&ms, e0, e1);
					best = std::min(best, ms);
				}
				printf("delay %3u run %4u B tps %2u: %.3f ms  %.0f GB/s\n", delay, run_bytes, tps, best,
				       total_elems * 4.0 / (best * 1e-3) / 1e9);
			}
		}
	}
	return 0;
}
