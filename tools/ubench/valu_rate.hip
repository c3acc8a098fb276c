// Micro-benchmark: issue rate of the integer VALU instructions the scatter kernel leans on.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench/valu_rate.hip -o /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define ITER 4096

template <int KIND>
__global__ __launch_bounds__(256) void k(unsigned *out, unsigned seed)
{
	unsigned a0 = threadIdx.x ^ seed, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19;
	for (int i = 0; i < ITER; ++i) {
		if (KIND == 0) {   // v_xor_b32
			asm volatile("v_xor_b32 %0, %0, %8\n v_xor_b32 %1, %1, %8\n v_xor_b32 %2, %2, %8\n v_xor_b32 %3, %3, %8\n"
			             "v_xor_b32 %4, %4, %8\n v_xor_b32 %5, %5, %8\n v_xor_b32 %6, %6, %8\n v_xor_b32 %7, %7, %8\n"
			             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(seed));
		} else if (KIND == 1) {   // v_fma_f32
			asm volatile("v_fma_f32 %0, %0, %8, %8\n v_fma_f32 %1, %1, %8, %8\n v_fma_f32 %2, %2, %8, %8\n v_fma_f32 %3, %3, %8, %8\n"
			             "v_fma_f32 %4, %4, %8, %8\n v_fma_f32 %5, %5, %8, %8\n v_fma_f32 %6, %6, %8, %8\n v_fma_f32 %7, %7, %8, %8\n"
			             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(seed));
		} else if (KIND == 2) {   // v_cmp_ne_u32 -> sgpr pair (VOP3)
			asm volatile("v_cmp_ne_u32 s[20:21], %0, %8\n v_cmp_ne_u32 s[22:23], %1, %8\n v_cmp_ne_u32 s[24:25], %2, %8\n v_cmp_ne_u32 s[26:27], %3, %8\n"
			             "v_cmp_ne_u32 s[28:29], %4, %8\n v_cmp_ne_u32 s[30:31], %5, %8\n v_cmp_ne_u32 s[32:33], %6, %8\n v_cmp_ne_u32 s[34:35], %7, %8\n"
			             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(seed)
			             : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "s28", "s29", "s30", "s31", "s32", "s33", "s34", "s35");
		} else if (KIND == 3) {   // v_bitop3_b32 (a & ~(b ^ c)) = 0x.. any code
			asm volatile("v_bitop3_b32 %0, %0, %8, %1 bitop3:0x90\n v_bitop3_b32 %1, %1, %8, %2 bitop3:0x90\n v_bitop3_b32 %2, %2, %8, %3 bitop3:0x90\n v_bitop3_b32 %3, %3, %8, %4 bitop3:0x90\n"
			             "v_bitop3_b32 %4, %4, %8, %5 bitop3:0x90\n v_bitop3_b32 %5, %5, %8, %6 bitop3:0x90\n v_bitop3_b32 %6, %6, %8, %7 bitop3:0x90\n v_bitop3_b32 %7, %7, %8, %0 bitop3:0x90\n"
			             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(seed));
		} else if (KIND == 4) {   // v_bfe_i32
			asm volatile("v_bfe_i32 %0, %0, 3, 1\n v_bfe_i32 %1, %1, 3, 1\n v_bfe_i32 %2, %2, 3, 1\n v_bfe_i32 %3, %3, 3, 1\n"
			             "v_bfe_i32 %4, %4, 3, 1\n v_bfe_i32 %5, %5, 3, 1\n v_bfe_i32 %6, %6, 3, 1\n v_bfe_i32 %7, %7, 3, 1\n"
			             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(seed));
		} else if (KIND == 5) {   // v_mbcnt_lo
			asm volatile("v_mbcnt_lo_u32_b32 %0, %0, %8\n v_mbcnt_lo_u32_b32 %1, %1, %8\n v_mbcnt_lo_u32_b32 %2, %2, %8\n v_mbcnt_lo_u32_b32 %3, %3, %8\n"
			             "v_mbcnt_lo_u32_b32 %4, %4, %8\n v_mbcnt_lo_u32_b32 %5, %5, %8\n v_mbcnt_lo_u32_b32 %6, %6, %8\n v_mbcnt_lo_u32_b32 %7, %7, %8\n"
			             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(seed));
		} else if (KIND == 6) {   // v_and_b32 VOP2 with sgpr operand
			asm volatile("v_and_b32 %0, s4, %0\n v_and_b32 %1, s4, %1\n v_and_b32 %2, s4, %2\n v_and_b32 %3, s4, %3\n"
			             "v_and_b32 %4, s4, %4\n v_and_b32 %5, s4, %5\n v_and_b32 %6, s4, %6\n v_and_b32 %7, s4, %7\n"
			             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(seed));
		} else if (KIND == 7) {   // v_lshl_add_u32 (VOP3, 3 operands)
			asm volatile("v_lshl_add_u32 %0, %0, 1, %8\n v_lshl_add_u32 %1, %1, 1, %8\n v_lshl_add_u32 %2, %2, 1, %8\n v_lshl_add_u32 %3, %3, 1, %8\n"
			             "v_lshl_add_u32 %4, %4, 1, %8\n v_lshl_add_u32 %5, %5, 1, %8\n v_lshl_add_u32 %6, %6, 1, %8\n v_lshl_add_u32 %7, %7, 1, %8\n"
			             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(seed));
		}
	}
	out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}

template <int KIND>
void run(const char *name, unsigned *d, int blocks_per_cu)
{
	const int blocks = 256 * blocks_per_cu;
	hipEvent_t e0, e1;
	hipEventCreate(&e0);
	hipEventCreate(&e1);
	hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, 1u);
	hipDeviceSynchronize();
	hipEventRecord(e0);
	hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, 1u);
	hipEventRecord(e1);
	hipEventSynchronize(e1);
	float ms;
	hipEventElapsedTime(&ms, e0, e1);
	// wave-instructions per SIMD: each block = 4 waves = 1 per SIMD; blocks_per_cu waves per SIMD
	const double insts_per_simd = (double)ITER * 8 * blocks_per_cu;
	const double ns_per_inst = ms * 1e6 / insts_per_simd;
	printf("%-16s waves/SIMD %d: %8.3f ms  %.3f ns per wave-inst per SIMD  (= %.2f cycles @2.4GHz)\n", name, blocks_per_cu, ms,
	       ns_per_inst, ns_per_inst * 2.4);
}

int main()
{
	unsigned *d;
	hipMalloc(&d, 256 * 8 * 256 * 4);
	for (int w : {1, 2, 4, 8}) {
		run<0>("v_xor_b32", d, w);
		run<1>("v_fma_f32", d, w);
		run<2>("v_cmp_ne->sgpr", d, w);
		run<3>("v_bitop3_b32", d, w);
		run<4>("v_bfe_i32", d, w);
		run<5>("v_mbcnt_lo", d, w);
		run<6>("v_and sgpr", d, w);
		run<7>("v_lshl_add_u32", d, w);
	}
	return 0;
}
