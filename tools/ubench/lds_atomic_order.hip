// Does ds_add_rtn_u32 hand out return values in lane order when several lanes of one wave-instruction
// hit the same LDS address?  (Undocumented; the stable ranking fast path would rely on it.)
// Each wave: rounds of `old = atomicAdd(&cnt[d], 1)`; the expected value is (count of d in earlier rounds)
// + (number of lower lanes with the same d in this round), computed independently with ballots.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef unsigned int u32;
typedef unsigned long long u64;

__device__ __forceinline__ u32 mbcnt64(u64 m) { return __builtin_amdgcn_mbcnt_hi((u32)(m >> 32), __builtin_amdgcn_mbcnt_lo((u32)m, 0u)); }

__device__ __forceinline__ u64 match8(u32 d)
{
	u64 m = ~0ull;
	for (int b = 0; b < 8; ++b) {
		const bool bit = (d >> b) & 1u;
		const u64 bal = __ballot(bit);
		m &= bit ? bal : ~bal;
	}
	return m;
}

template <int MODE>
__global__ __launch_bounds__(512) void k(u64 *bad, u32 seed, int rounds)
{
	__shared__ u32 cnt[8][256];
	__shared__ u32 ref[8][256];
	const u32 tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
	for (u32 i = tid; i < 8 * 256; i += 512) {
		(&cnt[0][0])[i] = 0;
		(&ref[0][0])[i] = 0;
	}
	__syncthreads();
	u32 x = seed ^ (blockIdx.x * 2654435761u) ^ (tid * 40503u);
	u64 nbad = 0;
	for (int r = 0; r < rounds; ++r) {
		x = x * 1664525u + 1013904223u;
		u32 d;
		if (MODE == 0) d = (x >> 13) & 0xFF;                 // uniform
		else if (MODE == 1) d = (x >> 13) & 1;               // two digits
		else if (MODE == 2) d = 7;                           // all equal
		else if (MODE == 3) d = ((x >> 13) & 7) * 32;        // same bank, 8 addresses
		else if (MODE == 4) d = (lane * 5 + r) & 0xFF & ~3u; // structured
		else d = ((x >> 9) % 3 == 0) ? 200 : ((x >> 13) & 0xFF); // one hot digit + uniform
		const u32 old = atomicAdd(&cnt[wid][d], 1u);
		const u64 m = match8(d);
		const u32 below = mbcnt64(m);
		const u32 prev = ref[wid][d];
		if (below == (u32)__popcll(m) - 1)
			ref[wid][d] = prev + (u32)__popcll(m);
		asm volatile("" ::: "memory");
		if (old != prev + below)
			++nbad;
	}
	if (nbad)
		atomicAdd(bad, nbad);
}

template <int MODE>
void run(const char *name, u64 *d_bad)
{
	hipMemset(d_bad, 0, 8);
	hipLaunchKernelGGL(k<MODE>, dim3(2048), dim3(512), 0, 0, d_bad, 12345u, 2000);
	hipDeviceSynchronize();
	u64 bad = 0;
	hipMemcpy(&bad, d_bad, 8, hipMemcpyDeviceToHost);
	printf("%-28s lanes checked %llu, out-of-lane-order returns %llu\n", name, 2048ull * 512 * 2000, bad);
}

int main()
{
	u64 *d_bad;
	hipMalloc(&d_bad, 8);
	run<0>("uniform 256", d_bad);
	run<1>("two digits", d_bad);
	run<2>("all equal", d_bad);
	run<3>("8 addresses on one bank", d_bad);
	run<4>("structured", d_bad);
	run<5>("hot digit + uniform", d_bad);
	return 0;
}
