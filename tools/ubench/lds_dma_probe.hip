// lds_dma_probe: what would the pass kernel gain from (a) LDS-DMA loads (global_load_lds_dwordx4: no key registers in the
// loading waves) and (b) a role split -- waves that only load and waves that only write out, each with its own vmcnt stream
// (round 2's review, item 1b)?  Answered on the pass's DATA MOVEMENT alone: a tile goes global -> LDS ->
// global, nothing is ranked.  Whatever the ranking costs comes on top, so these rates bound what either change can give.
//
//   direct      global -> registers -> global (the plain copy: 6.3 TB/s on this part)
//   lds_regs    one 32 Ki-key tile per workgroup (the pass's shape): 16-byte loads -> ds_write_b128 -> barrier -> ds_read_b128 -> stores
//   lds_dma     the same tile fetched with LDS-DMA: global_load_lds_dwordx4 -> vmcnt(0) -> barrier -> ds_read_b128 -> stores
//   split_dma   persistent workgroups, 8 Ki-key tiles in four LDS buffers: waves 0-7 only issue LDS-DMA (three tiles ahead),
//               waves 8-15 only read the LDS and store; one s_barrier per tile; the loaders wait on vmcnt(8) (their own DMAs),
//               the writers never wait for their stores
//   split_regs  the same roles with register loads in the loaders (global_load -> ds_write)
// Each with the output written linearly and as 512-byte runs at permuted places (what a pass's write-out looks like when
// every run starts on a 512-byte boundary).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench/lds_dma_probe.hip -o tools/ubench/lds_dma_probe.bin
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

typedef uint32_t u32;
typedef uint64_t u64;
typedef u32 u32x4 __attribute__((ext_vector_type(4)));

#define CK(x)                                                                         \
	do {                                                                              \
		hipError_t e_ = (x);                                                          \
		if (e_ != hipSuccess) {                                                       \
			printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
			exit(1);                                                                  \
		}                                                                             \
	} while (0)

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

// run r (128 keys = 512 bytes) of the input goes to run perm(r) of the output: odd multiplier modulo a power of two
__device__ __forceinline__ u64 out_vec_index(u64 v, u32 scatter, u64 nruns_mask)   // v: index of a 16-byte vector
{
	if (!scatter)
		return v;
	const u64 run = v >> 5, in = v & 31;   // 32 vectors per run
	return (((run * 0x9E3779B1ull) & nruns_mask) << 5) | in;
}

__global__ __launch_bounds__(1024) void direct_kernel(const u32x4 *__restrict__ in, u32x4 *__restrict__ out, u64 nvec, u32 scatter,
                                                      u64 nruns_mask)
{
	const u64 base = (u64)blockIdx.x * 8192;   // a tile of 32 Ki keys = 8192 vectors
	u32x4 v[8];
#pragma unroll
	for (int i = 0; i < 8; ++i)
		v[i] = in[base + i * 1024 + threadIdx.x];
#pragma unroll
	for (int i = 0; i < 8; ++i)
		out[out_vec_index(base + i * 1024 + threadIdx.x, scatter, nruns_mask)] = v[i];
}

__global__ __launch_bounds__(1024) void lds_regs_kernel(const u32x4 *__restrict__ in, u32x4 *__restrict__ out, u64 nvec, u32 scatter,
                                                        u64 nruns_mask)
{
	extern __shared__ u32x4 lds[];   // 8192 vectors = 128 KiB
	const u64 base = (u64)blockIdx.x * 8192;
	const u32 tid = threadIdx.x;
	u32x4 v[8];
#pragma unroll
	for (int i = 0; i < 8; ++i)
		v[i] = in[base + i * 1024 + tid];
#pragma unroll
	for (int i = 0; i < 8; ++i)
		lds[i * 1024 + tid] = v[i];
	__syncthreads();
	const u32 t2 = (tid + 320) & 1023;   // (another wave's data)
#pragma unroll
	for (int i = 0; i < 8; ++i)
		v[i] = lds[i * 1024 + t2];
#pragma unroll
	for (int i = 0; i < 8; ++i)
		out[out_vec_index(base + i * 1024 + t2, scatter, nruns_mask)] = v[i];
}

__device__ __forceinline__ void dma16(const u32x4 *g, u32x4 *l)   // lane's source, the WAVE's destination base (+ lane * 16 by hardware)
{
	__builtin_amdgcn_global_load_lds((glb_void *)g, (lds_void *)l, 16, 0, 0);
}

__global__ __launch_bounds__(1024) void lds_dma_kernel(const u32x4 *__restrict__ in, u32x4 *__restrict__ out, u64 nvec, u32 scatter,
                                                       u64 nruns_mask)
{
	extern __shared__ u32x4 lds[];
	const u64 base = (u64)blockIdx.x * 8192;
	const u32 tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
#pragma unroll
	for (int i = 0; i < 8; ++i)
		dma16(in + base + i * 1024 + wid * 64 + lane, lds + i * 1024 + wid * 64);
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
	__builtin_amdgcn_s_barrier();
	const u32 t2 = (tid + 320) & 1023;
	u32x4 v[8];
#pragma unroll
	for (int i = 0; i < 8; ++i)
		v[i] = lds[i * 1024 + t2];
#pragma unroll
	for (int i = 0; i < 8; ++i)
		out[out_vec_index(base + i * 1024 + t2, scatter, nruns_mask)] = v[i];
}

// Role split.  Tile = 2048 vectors (8 Ki keys, 32 KiB), NB buffers.  Workgroup b takes tiles b, b + G, b + 2G, ...
template <bool DMA>
__global__ __launch_bounds__(1024) void split_kernel(const u32x4 *__restrict__ in, u32x4 *__restrict__ out, u64 ntiles, u32 scatter,
                                                     u64 nruns_mask)
{
	constexpr u32 TV = 2048, NB = 4, AHEAD = 3;
	extern __shared__ u32x4 lds[];   // NB * TV vectors = 128 KiB
	const u32 tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
	const u64 G = gridDim.x;
	const u64 mine = ntiles > blockIdx.x ? (ntiles - blockIdx.x + G - 1) / G : 0;   // tiles of this workgroup
	const bool loader = wid < 8;
	if (loader) {
		// 4 instructions of 64 lanes x 16 bytes per tile and wave
		auto issue = [&](u64 k) {
			const u64 t = blockIdx.x + k * G;
			const u32x4 *g = in + t * TV + wid * 256 + lane;
			u32x4 *l = lds + (k % NB) * TV + wid * 256;
#pragma unroll
			for (int i = 0; i < 4; ++i)
				dma16(g + i * 64, l + i * 64);
		};
		if constexpr (DMA) {
			for (u64 k = 0; k < AHEAD && k < mine; ++k)
				issue(k);
			for (u64 k = 0; k < mine; ++k) {
				// tile k has landed when at most the DMAs of the (up to two) later tiles are outstanding
				const u64 later = mine - 1 - k < AHEAD - 1 ? mine - 1 - k : AHEAD - 1;
				if (later == 2)
					asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
				else if (later == 1)
					asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
				else
					asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
				__builtin_amdgcn_s_barrier();   // A(k): tile k is in the LDS; the writers are through with buffer (k - 1) % NB
				if (k + AHEAD < mine)
					issue(k + AHEAD);           // into buffer (k + 3) % 4 == (k - 1) % 4
			}
		} else {
			// register loads: tile k + 1 is requested before tile k is written into the LDS (the compiler counts the waits)
			u32x4 va[4], vb[4];
			auto request = [&](u32x4 (&v)[4], u64 k) {
				const u32x4 *g = in + (blockIdx.x + k * G) * TV + wid * 256 + lane;
#pragma unroll
				for (int i = 0; i < 4; ++i)
					v[i] = g[i * 64];
			};
			if (mine)
				request(va, 0);
			for (u64 k = 0; k < mine; ++k) {
				if (k + 1 < mine)
					request(vb, k + 1);
				u32x4 *l = lds + (k % NB) * TV + wid * 256 + lane;
#pragma unroll
				for (int i = 0; i < 4; ++i)
					l[i * 64] = va[i];
				asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
				__builtin_amdgcn_s_barrier();
#pragma unroll
				for (int i = 0; i < 4; ++i)
					va[i] = vb[i];
			}
		}
	} else {
		const u32 w = wid - 8;
		for (u64 k = 0; k < mine; ++k) {
			__builtin_amdgcn_s_barrier();       // A(k)
			const u64 t = blockIdx.x + k * G;
			const u32x4 *l = lds + (k % NB) * TV + ((w + 3) & 7) * 256 + lane;
			u32x4 v[4];
#pragma unroll
			for (int i = 0; i < 4; ++i)
				v[i] = l[i * 64];
			asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
			for (int i = 0; i < 4; ++i)
				out[out_vec_index(t * TV + ((w + 3) & 7) * 256 + i * 64 + lane, scatter, nruns_mask)] = v[i];
		}
	}
}

__global__ void gen_kernel(u32 *dst, u64 n)
{
	for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x)
		dst[i] = (u32)(i * 2654435761u) ^ (u32)(i >> 13);
}

// the output must hold the input's vectors at their (permuted) places
__global__ void check_kernel(const u32x4 *in, const u32x4 *out, u64 nvec, u32 scatter, u64 nruns_mask, u64 *bad)
{
	u64 b = 0;
	for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (u64)gridDim.x * blockDim.x) {
		const u32x4 a = in[i], c = out[out_vec_index(i, scatter, nruns_mask)];
		b += (a.x != c.x) | (a.y != c.y) | (a.z != c.z) | (a.w != c.w);
	}
	if (b)
		atomicAdd((unsigned long long *)bad, (unsigned long long)b);
}

int main(int argc, char **argv)
{
	const int log2n = argc > 1 ? atoi(argv[1]) : 28;
	const u64 n = 1ull << log2n, nvec = n / 4, nruns_mask = n / 128 - 1;
	u32 *in, *out;
	u64 *bad;
	CK(hipMalloc(&in, n * 4));
	CK(hipMalloc(&out, n * 4));
	CK(hipMalloc(&bad, 8));
	hipLaunchKernelGGL(gen_kernel, dim3(4096), dim3(256), 0, 0, in, n);
	CK(hipDeviceSynchronize());
	CK(hipFuncSetAttribute((const void *)lds_regs_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
	CK(hipFuncSetAttribute((const void *)lds_dma_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
	CK(hipFuncSetAttribute((const void *)split_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
	CK(hipFuncSetAttribute((const void *)split_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
	hipEvent_t e0, e1;
	CK(hipEventCreate(&e0));
	CK(hipEventCreate(&e1));
	const unsigned tiles32 = (unsigned)(n >> 15);
	const u64 tiles8 = n >> 13;
	printf("%d GiB in, as much out; u32 keys 2^%d; rates count both directions (what DESIGN.md calls algorithmic bytes)\n",
	       (int)((n * 4) >> 30), log2n);
	for (u32 scatter = 0; scatter < 2; ++scatter) {
		for (int variant = 0; variant < 6; ++variant) {
			const char *names[6] = {"direct", "lds_regs", "lds_dma", "split_dma  (256 workgroups)", "split_regs (256 workgroups)",
			                        "split_dma  (a workgroup per 512 tiles)"};
			float best = 1e9f;
			for (int rep = 0; rep < 7; ++rep) {
				if (rep == 0)
					CK(hipMemset(out, 0, n * 4));
				CK(hipEventRecord(e0, 0));
				switch (variant) {
				case 0:
					hipLaunchKernelGGL(direct_kernel, dim3(tiles32), dim3(1024), 0, 0, (const u32x4 *)in, (u32x4 *)out, nvec, scatter, nruns_mask);
					break;
				case 1:
					hipLaunchKernelGGL(lds_regs_kernel, dim3(tiles32), dim3(1024), 131072, 0, (const u32x4 *)in, (u32x4 *)out, nvec, scatter, nruns_mask);
					break;
				case 2:
					hipLaunchKernelGGL(lds_dma_kernel, dim3(tiles32), dim3(1024), 131072, 0, (const u32x4 *)in, (u32x4 *)out, nvec, scatter, nruns_mask);
					break;
				case 3:
					hipLaunchKernelGGL(split_kernel<true>, dim3(256), dim3(1024), 131072, 0, (const u32x4 *)in, (u32x4 *)out, tiles8, scatter, nruns_mask);
					break;
				case 4:
					hipLaunchKernelGGL(split_kernel<false>, dim3(256), dim3(1024), 131072, 0, (const u32x4 *)in, (u32x4 *)out, tiles8, scatter, nruns_mask);
					break;
				case 5:
					hipLaunchKernelGGL(split_kernel<true>, dim3((unsigned)((tiles8 + 511) / 512)), dim3(1024), 131072, 0, (const u32x4 *)in, (u32x4 *)out, tiles8, scatter, nruns_mask);
					break;
				}
				CK(hipEventRecord(e1, 0));
				CK(hipEventSynchronize(e1));
				CK(hipGetLastError());
				float ms;
				CK(hipEventElapsedTime(&ms, e0, e1));
				if (rep == 0) {
					CK(hipMemset(bad, 0, 8));
					hipLaunchKernelGGL(check_kernel, dim3(4096), dim3(256), 0, 0, (const u32x4 *)in, (const u32x4 *)out, nvec, scatter, nruns_mask, bad);
					u64 hb;
					CK(hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost));
					if (hb) {
						printf("%-44s WRONG: %llu vectors differ\n", names[variant], (unsigned long long)hb);
						break;
					}
				} else if (ms < best) {
					best = ms;
				}
			}
			if (best < 1e8f)
				printf("%-10s %-44s %.3f ms  %.2f TB/s\n", scatter ? "512 B runs" : "linear", names[variant], best, 2.0 * n * 4 / best / 1e9);
		}
	}
	return 0;
}
