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

// Touch-first variant: before a tile's runs are written, the 128-byte lines its runs start and end in are READ (one lane
// per line), so that the partial writes find their lines in the Infinity Cache.
__global__ __launch_bounds__(512) void kt(u32 *out, u64 stream_stride_elems, u32 run_elems, u32 ntiles, u32 misalign, u32 *ticket,
                                          u32 *sink)
{
	__shared__ u32 s_t;
	const u32 tid = threadIdx.x;
	u32 acc = 0;
	for (;;) {
		if (tid == 0)
			s_t = atomicAdd(ticket, 1u);
		__syncthreads();
		const u32 t = s_t;
		__syncthreads();
		if (t >= ntiles)
			break;
		{
			// 256 runs x 2 ends = 512 lines, one per thread
			const u32 r = tid >> 1, endp = tid & 1;
			const u32 *p = out + (u64)r * stream_stride_elems + misalign * (r & 3) + (u64)t * run_elems + (endp ? run_elems - 1 : 0);
			acc += __builtin_nontemporal_load(p);
		}
		const u32 tile_elems = 256 * run_elems;
		for (u32 i0 = tid * 4; i0 < tile_elems; i0 += 512 * 4) {
			const u32 r = i0 / run_elems, o = i0 % run_elems;
			u32 *dst = out + (u64)r * stream_stride_elems + misalign * (r & 3) + (u64)t * run_elems + o;
			*(uu32x4 *)dst = u32x4{i0, t, r, o};
		}
	}
	if (acc == 0x12345678u)
		sink[0] = acc;
}

// XCD-affine variant: workgroup b (on XCD b % 8) takes its tiles from XCD b % 8's own counter; the s-th ticket of XCD x
// is tile (s / K) * 8K + x * K + s % K: blocks of K consecutive tiles stay on one XCD (one L2).
__global__ __launch_bounds__(512) void kx(u32 *out, u64 stream_stride_elems, u32 run_elems, u32 ntiles, u32 misalign, u32 *ticket,
                                          u32 K)
{
	__shared__ u32 s_t;
	const u32 tid = threadIdx.x, x = blockIdx.x & 7;
	for (;;) {
		if (tid == 0)
			s_t = atomicAdd(ticket + x * 32, 1u);
		__syncthreads();
		const u32 s = s_t;
		__syncthreads();
		const u32 t = (s / K) * 8 * K + x * K + s % K;
		if (t >= ntiles)
			return;
		const u32 tile_elems = 256 * run_elems;
		for (u32 i0 = tid * 4; i0 < tile_elems; i0 += 512 * 4) {
			const u32 r = i0 / run_elems, o = i0 % run_elems;
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
	hipMalloc(&d_ticket, 1024);
	printf("-- touch the boundary lines first (misaligned 512-B runs, 1 GiB)\n");
	for (int variant = 0; variant < 2; ++variant) {
		const u32 run_elems = 128;
		const u32 ntiles = (u32)(total_elems / (256ull * run_elems));
		const u64 stride = total_elems / 256;
		float best = 1e9;
		for (int rep = 0; rep < 3; ++rep) {
			(void)hipMemset(d_ticket, 0, 1024);
			hipEvent_t e0, e1;
			(void)hipEventCreate(&e0);
			(void)hipEventCreate(&e1);
			(void)hipEventRecord(e0);
			if (variant == 0)
				hipLaunchKernelGGL(k, dim3(512), dim3(512), 0, 0, d, stride, run_elems, ntiles, 1u, d_ticket);
			else
				hipLaunchKernelGGL(kt, dim3(512), dim3(512), 0, 0, d, stride, run_elems, ntiles, 1u, d_ticket, d_ticket + 128);
			(void)hipEventRecord(e1);
			(void)hipEventSynchronize(e1);
			float ms;
			(void)hipEventElapsedTime(&ms, e0, e1);
			best = std::min(best, ms);
		}
		printf("%s: %.3f ms  %.0f GB/s\n", variant ? "touch first" : "plain      ", best, total_elems * 4.0 / (best * 1e-3) / 1e9);
	}
	printf("-- working-set size: misaligned 512-B runs into a region of W bytes, rewritten until 4 GiB are stored (does the Infinity Cache absorb the partial writes?)\n");
	for (u64 wlog : {24ull, 25ull, 26ull, 27ull, 28ull, 30ull}) {
		const u64 welems = (1ull << wlog) / 4;
		const u32 run_elems = 128;
		const u32 ntiles = (u32)(welems / (256ull * run_elems));
		const u64 stride = welems / 256;
		const int reps = (int)((1ull << 32) / (1ull << wlog));
		for (u32 m : {0u, 1u}) {
			hipEvent_t e0, e1;
			(void)hipEventCreate(&e0);
			(void)hipEventCreate(&e1);
			(void)hipMemset(d_ticket, 0, 1024);
			hipLaunchKernelGGL(k, dim3(512), dim3(512), 0, 0, d, stride, run_elems, ntiles, m, d_ticket);
			(void)hipDeviceSynchronize();
			float tot = 0;
			for (int rep = 0; rep < reps; ++rep) {
				(void)hipMemset(d_ticket, 0, 1024);
				(void)hipEventRecord(e0);
				hipLaunchKernelGGL(k, dim3(512), dim3(512), 0, 0, d, stride, run_elems, ntiles, m, d_ticket);
				(void)hipEventRecord(e1);
				(void)hipEventSynchronize(e1);
				float ms;
				(void)hipEventElapsedTime(&ms, e0, e1);
				tot += ms;
			}
			printf("W = 2^%llu B shift %u: %.0f GB/s\n", wlog, m, (double)reps * welems * 4.0 / (tot * 1e-3) / 1e9);
		}
	}
	printf("-- stream r shifted by m * (r & 3) elements: which misalignment costs?\n");
	for (u32 m : {0u, 1u, 2u, 4u, 8u, 16u, 32u}) {
		const u32 run_bytes = 512, run_elems = run_bytes / 4;
		const u32 ntiles = (u32)(total_elems / (256ull * run_elems));
		const u64 stride = total_elems / 256;
		float best = 1e9;
		for (int rep = 0; rep < 3; ++rep) {
			(void)hipMemset(d_ticket, 0, 1024);
			hipEvent_t e0, e1;
			(void)hipEventCreate(&e0);
			(void)hipEventCreate(&e1);
			(void)hipEventRecord(e0);
			hipLaunchKernelGGL(k, dim3(512), dim3(512), 0, 0, d, stride, run_elems, ntiles, m, d_ticket);
			(void)hipEventRecord(e1);
			(void)hipEventSynchronize(e1);
			float ms;
			(void)hipEventElapsedTime(&ms, e0, e1);
			best = std::min(best, ms);
		}
		printf("run 512 B shift %2u elements: %.3f ms  %.0f GB/s\n", m, best, total_elems * 4.0 / (best * 1e-3) / 1e9);
	}
	printf("-- XCD-affine tile blocks (misaligned streams)\n");
	for (u32 run_bytes : {128u, 256u, 512u, 1024u}) {
		for (u32 K : {1u, 4u, 32u, 128u}) {
			const u32 run_elems = run_bytes / 4;
			const u32 ntiles = (u32)(total_elems / (256ull * run_elems));
			const u64 stride = total_elems / 256;
			float best = 1e9;
			for (int rep = 0; rep < 3; ++rep) {
				hipMemset(d_ticket, 0, 1024);
				hipEvent_t e0, e1;
				hipEventCreate(&e0);
				hipEventCreate(&e1);
				hipEventRecord(e0);
				hipLaunchKernelGGL(kx, dim3(512), dim3(512), 0, 0, d, stride, run_elems, ntiles, 1u, d_ticket, K);
				hipEventRecord(e1);
				hipEventSynchronize(e1);
				float ms;
				hipEventElapsedTime(&ms, e0, e1);
				best = std::min(best, ms);
			}
			printf("run %5u B  K %3u: %.3f ms  %.0f GB/s\n", run_bytes, K, best, total_elems * 4.0 / (best * 1e-3) / 1e9);
		}
	}
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
