// Micro-benchmark 2: issue cost (cycles per wave-instruction per SIMD) of candidate instructions for the
// wave-ranking code, at 8 waves/SIMD.  Each kernel runs 8 independent chains.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define ITER 32768
#define OP8(fmt) \
	fmt(0) fmt(1) fmt(2) fmt(3) fmt(4) fmt(5) fmt(6) fmt(7)

#define DEFK(NAME, BODY, CLOB...)                                                                              \
	__global__ __launch_bounds__(256) void NAME(unsigned *out, unsigned seed)                                  \
	{                                                                                                          \
		unsigned a0 = threadIdx.x ^ seed, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13,   \
		         a6 = a0 * 17, a7 = a0 * 19;                                                                   \
		unsigned b = seed * 77;                                                                                \
		for (int i = 0; i < ITER; ++i) {                                                                       \
			asm volatile(BODY                                                                                  \
			             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)      \
			             : "v"(seed), "v"(b)                                                                   \
			             : CLOB);                                                                              \
		}                                                                                                      \
		out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;                    \
	}

DEFK(k_xor, "v_xor_b32 %0, %0, %8\n v_xor_b32 %1, %1, %8\n v_xor_b32 %2, %2, %8\n v_xor_b32 %3, %3, %8\n v_xor_b32 %4, %4, %8\n v_xor_b32 %5, %5, %8\n v_xor_b32 %6, %6, %8\n v_xor_b32 %7, %7, %8\n", "memory")
DEFK(k_and, "v_and_b32 %0, %0, %8\n v_and_b32 %1, %1, %8\n v_and_b32 %2, %2, %8\n v_and_b32 %3, %3, %8\n v_and_b32 %4, %4, %8\n v_and_b32 %5, %5, %8\n v_and_b32 %6, %6, %8\n v_and_b32 %7, %7, %8\n", "memory")
DEFK(k_and_s, "v_and_b32 %0, s4, %0\n v_and_b32 %1, s4, %1\n v_and_b32 %2, s4, %2\n v_and_b32 %3, s4, %3\n v_and_b32 %4, s4, %4\n v_and_b32 %5, s4, %5\n v_and_b32 %6, s4, %6\n v_and_b32 %7, s4, %7\n", "memory")
DEFK(k_and_c, "v_and_b32 %0, 7, %0\n v_and_b32 %1, 7, %1\n v_and_b32 %2, 7, %2\n v_and_b32 %3, 7, %3\n v_and_b32 %4, 7, %4\n v_and_b32 %5, 7, %5\n v_and_b32 %6, 7, %6\n v_and_b32 %7, 7, %7\n", "memory")
DEFK(k_add, "v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8\n", "memory")
DEFK(k_mov, "v_mov_b32 %0, %8\n v_mov_b32 %1, %8\n v_mov_b32 %2, %8\n v_mov_b32 %3, %8\n v_mov_b32 %4, %8\n v_mov_b32 %5, %8\n v_mov_b32 %6, %8\n v_mov_b32 %7, %8\n", "memory")
DEFK(k_lshl, "v_lshlrev_b32 %0, 1, %0\n v_lshlrev_b32 %1, 1, %1\n v_lshlrev_b32 %2, 1, %2\n v_lshlrev_b32 %3, 1, %3\n v_lshlrev_b32 %4, 1, %4\n v_lshlrev_b32 %5, 1, %5\n v_lshlrev_b32 %6, 1, %6\n v_lshlrev_b32 %7, 1, %7\n", "memory")
DEFK(k_lshl_v, "v_lshlrev_b32 %0, %8, %0\n v_lshlrev_b32 %1, %8, %1\n v_lshlrev_b32 %2, %8, %2\n v_lshlrev_b32 %3, %8, %3\n v_lshlrev_b32 %4, %8, %4\n v_lshlrev_b32 %5, %8, %5\n v_lshlrev_b32 %6, %8, %6\n v_lshlrev_b32 %7, %8, %7\n", "memory")
DEFK(k_bitop3, "v_bitop3_b32 %0, %0, %8, %9 bitop3:0x90\n v_bitop3_b32 %1, %1, %8, %9 bitop3:0x90\n v_bitop3_b32 %2, %2, %8, %9 bitop3:0x90\n v_bitop3_b32 %3, %3, %8, %9 bitop3:0x90\n v_bitop3_b32 %4, %4, %8, %9 bitop3:0x90\n v_bitop3_b32 %5, %5, %8, %9 bitop3:0x90\n v_bitop3_b32 %6, %6, %8, %9 bitop3:0x90\n v_bitop3_b32 %7, %7, %8, %9 bitop3:0x90\n", "memory")
DEFK(k_bitop3_s, "v_bitop3_b32 %0, %0, s4, %9 bitop3:0x90\n v_bitop3_b32 %1, %1, s4, %9 bitop3:0x90\n v_bitop3_b32 %2, %2, s4, %9 bitop3:0x90\n v_bitop3_b32 %3, %3, s4, %9 bitop3:0x90\n v_bitop3_b32 %4, %4, s4, %9 bitop3:0x90\n v_bitop3_b32 %5, %5, s4, %9 bitop3:0x90\n v_bitop3_b32 %6, %6, s4, %9 bitop3:0x90\n v_bitop3_b32 %7, %7, s4, %9 bitop3:0x90\n", "memory")
DEFK(k_and_or, "v_and_or_b32 %0, %0, %8, %9\n v_and_or_b32 %1, %1, %8, %9\n v_and_or_b32 %2, %2, %8, %9\n v_and_or_b32 %3, %3, %8, %9\n v_and_or_b32 %4, %4, %8, %9\n v_and_or_b32 %5, %5, %8, %9\n v_and_or_b32 %6, %6, %8, %9\n v_and_or_b32 %7, %7, %8, %9\n", "memory")
DEFK(k_add3, "v_add3_u32 %0, %0, %8, %9\n v_add3_u32 %1, %1, %8, %9\n v_add3_u32 %2, %2, %8, %9\n v_add3_u32 %3, %3, %8, %9\n v_add3_u32 %4, %4, %8, %9\n v_add3_u32 %5, %5, %8, %9\n v_add3_u32 %6, %6, %8, %9\n v_add3_u32 %7, %7, %8, %9\n", "memory")
DEFK(k_lshl_add, "v_lshl_add_u32 %0, %0, 1, %8\n v_lshl_add_u32 %1, %1, 1, %8\n v_lshl_add_u32 %2, %2, 1, %8\n v_lshl_add_u32 %3, %3, 1, %8\n v_lshl_add_u32 %4, %4, 1, %8\n v_lshl_add_u32 %5, %5, 1, %8\n v_lshl_add_u32 %6, %6, 1, %8\n v_lshl_add_u32 %7, %7, 1, %8\n", "memory")
DEFK(k_lshl_add_v, "v_lshl_add_u32 %0, %0, %9, %8\n v_lshl_add_u32 %1, %1, %9, %8\n v_lshl_add_u32 %2, %2, %9, %8\n v_lshl_add_u32 %3, %3, %9, %8\n v_lshl_add_u32 %4, %4, %9, %8\n v_lshl_add_u32 %5, %5, %9, %8\n v_lshl_add_u32 %6, %6, %9, %8\n v_lshl_add_u32 %7, %7, %9, %8\n", "memory")
DEFK(k_bfe_i, "v_bfe_i32 %0, %0, 3, 1\n v_bfe_i32 %1, %1, 3, 1\n v_bfe_i32 %2, %2, 3, 1\n v_bfe_i32 %3, %3, 3, 1\n v_bfe_i32 %4, %4, 3, 1\n v_bfe_i32 %5, %5, 3, 1\n v_bfe_i32 %6, %6, 3, 1\n v_bfe_i32 %7, %7, 3, 1\n", "memory")
DEFK(k_bfe_v, "v_bfe_u32 %0, %0, %8, %9\n v_bfe_u32 %1, %1, %8, %9\n v_bfe_u32 %2, %2, %8, %9\n v_bfe_u32 %3, %3, %8, %9\n v_bfe_u32 %4, %4, %8, %9\n v_bfe_u32 %5, %5, %8, %9\n v_bfe_u32 %6, %6, %8, %9\n v_bfe_u32 %7, %7, %8, %9\n", "memory")
DEFK(k_fma, "v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n", "memory")
DEFK(k_fmac, "v_fmac_f32 %0, %8, %9\n v_fmac_f32 %1, %8, %9\n v_fmac_f32 %2, %8, %9\n v_fmac_f32 %3, %8, %9\n v_fmac_f32 %4, %8, %9\n v_fmac_f32 %5, %8, %9\n v_fmac_f32 %6, %8, %9\n v_fmac_f32 %7, %8, %9\n", "memory")
DEFK(k_cmp_vcc, "v_cmp_ne_u32 vcc, %0, %8\n v_cmp_ne_u32 vcc, %1, %8\n v_cmp_ne_u32 vcc, %2, %8\n v_cmp_ne_u32 vcc, %3, %8\n v_cmp_ne_u32 vcc, %4, %8\n v_cmp_ne_u32 vcc, %5, %8\n v_cmp_ne_u32 vcc, %6, %8\n v_cmp_ne_u32 vcc, %7, %8\n", "memory", "vcc")
DEFK(k_cmp_s, "v_cmp_ne_u32 s[20:21], %0, %8\n v_cmp_ne_u32 s[22:23], %1, %8\n v_cmp_ne_u32 s[24:25], %2, %8\n v_cmp_ne_u32 s[26:27], %3, %8\n v_cmp_ne_u32 s[28:29], %4, %8\n v_cmp_ne_u32 s[30:31], %5, %8\n v_cmp_ne_u32 s[32:33], %6, %8\n v_cmp_ne_u32 s[34:35], %7, %8\n", "memory", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "s28", "s29", "s30", "s31", "s32", "s33", "s34", "s35")
DEFK(k_cndmask, "v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc\n", "memory")
DEFK(k_bcnt, "v_bcnt_u32_b32 %0, %0, %8\n v_bcnt_u32_b32 %1, %1, %8\n v_bcnt_u32_b32 %2, %2, %8\n v_bcnt_u32_b32 %3, %3, %8\n v_bcnt_u32_b32 %4, %4, %8\n v_bcnt_u32_b32 %5, %5, %8\n v_bcnt_u32_b32 %6, %6, %8\n v_bcnt_u32_b32 %7, %7, %8\n", "memory")
DEFK(k_mbcnt_lo, "v_mbcnt_lo_u32_b32 %0, %0, %8\n v_mbcnt_lo_u32_b32 %1, %1, %8\n v_mbcnt_lo_u32_b32 %2, %2, %8\n v_mbcnt_lo_u32_b32 %3, %3, %8\n v_mbcnt_lo_u32_b32 %4, %4, %8\n v_mbcnt_lo_u32_b32 %5, %5, %8\n v_mbcnt_lo_u32_b32 %6, %6, %8\n v_mbcnt_lo_u32_b32 %7, %7, %8\n", "memory")
DEFK(k_xnor, "v_xnor_b32 %0, %0, %8\n v_xnor_b32 %1, %1, %8\n v_xnor_b32 %2, %2, %8\n v_xnor_b32 %3, %3, %8\n v_xnor_b32 %4, %4, %8\n v_xnor_b32 %5, %5, %8\n v_xnor_b32 %6, %6, %8\n v_xnor_b32 %7, %7, %8\n", "memory")
DEFK(k_perm, "v_perm_b32 %0, %0, %8, %9\n v_perm_b32 %1, %1, %8, %9\n v_perm_b32 %2, %2, %8, %9\n v_perm_b32 %3, %3, %8, %9\n v_perm_b32 %4, %4, %8, %9\n v_perm_b32 %5, %5, %8, %9\n v_perm_b32 %6, %6, %8, %9\n v_perm_b32 %7, %7, %8, %9\n", "memory")
DEFK(k_salu, "s_and_b64 s[20:21], s[20:21], s[22:23]\n s_and_b64 s[24:25], s[24:25], s[22:23]\n s_and_b64 s[26:27], s[26:27], s[22:23]\n s_and_b64 s[28:29], s[28:29], s[22:23]\n s_and_b64 s[30:31], s[30:31], s[22:23]\n s_and_b64 s[32:33], s[32:33], s[22:23]\n s_and_b64 s[34:35], s[34:35], s[22:23]\n s_and_b64 s[36:37], s[36:37], s[22:23]\n", "memory", "scc", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "s28", "s29", "s30", "s31", "s32", "s33", "s34", "s35", "s36", "s37")
// mixed: 4 VALU (xor) interleaved with 4 SALU
DEFK(k_mix_vs, "v_xor_b32 %0, %0, %8\n s_and_b64 s[20:21], s[20:21], s[22:23]\n v_xor_b32 %1, %1, %8\n s_and_b64 s[24:25], s[24:25], s[22:23]\n v_xor_b32 %2, %2, %8\n s_and_b64 s[26:27], s[26:27], s[22:23]\n v_xor_b32 %3, %3, %8\n s_and_b64 s[28:29], s[28:29], s[22:23]\n", "memory", "scc", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "s28", "s29")

typedef void (*kern_t)(unsigned *, unsigned);

void run(const char *name, kern_t kern, unsigned *d, int blocks_per_cu, int insts_per_iter)
{
	const int blocks = 256 * blocks_per_cu;
	hipEvent_t e0, e1;
	(void)hipEventCreate(&e0);
	(void)hipEventCreate(&e1);
	hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, 1u);
	(void)hipDeviceSynchronize();
	(void)hipEventRecord(e0);
	hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, 1u);
	(void)hipEventRecord(e1);
	(void)hipEventSynchronize(e1);
	float ms;
	(void)hipEventElapsedTime(&ms, e0, e1);
	const double insts_per_simd = (double)ITER * insts_per_iter * blocks_per_cu;
	const double ns_per_inst = ms * 1e6 / insts_per_simd;
	printf("%-14s w/SIMD %d: %7.3f ms  %.3f ns/inst/SIMD\n", name, blocks_per_cu, ms, ns_per_inst);
}

int main()
{
	unsigned *d;
	(void)hipMalloc(&d, 256 * 8 * 256 * 4);
#define R(k) run(#k, k, d, w, 8)
	for (int w : {2, 8}) {
		R(k_xor); R(k_and); R(k_and_s); R(k_and_c); R(k_add); R(k_mov); R(k_lshl); R(k_lshl_v); R(k_bitop3); R(k_bitop3_s);
		R(k_and_or); R(k_add3); R(k_lshl_add); R(k_lshl_add_v); R(k_bfe_i); R(k_bfe_v); R(k_fma); R(k_fmac); R(k_cmp_vcc);
		R(k_cmp_s); R(k_cndmask); R(k_bcnt); R(k_mbcnt_lo); R(k_xnor); R(k_perm); R(k_salu); R(k_mix_vs);
	}
	return 0;
}
