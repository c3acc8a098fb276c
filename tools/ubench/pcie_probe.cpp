// pcie_probe: what the host-pointer entry can hope for -- hipMemcpy of pageable and pinned memory, and how fast T threads
// copy pageable memory into a pinned bounce buffer.
// Build: hipcc -O2 -std=c++17 tools/ubench/pcie_probe.cpp -o tools/ubench/pcie_probe.bin -lpthread
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

static double now()
{
	return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main()
{
	const size_t bytes = 1ull << 30;
	char *pageable = (char *)malloc(bytes), *pinned = nullptr;
	void *dev = nullptr;
	memset(pageable, 1, bytes);
	(void)hipHostMalloc((void **)&pinned, bytes, hipHostMallocDefault);
	memset(pinned, 2, bytes);
	(void)hipMalloc(&dev, bytes);
	for (int rep = 0; rep < 2; ++rep) {
		double t = now();
		(void)hipMemcpy(dev, pageable, bytes, hipMemcpyHostToDevice);
		printf("H2D pageable: %.1f ms  %.1f GB/s\n", (now() - t) * 1e3, bytes / (now() - t) / 1e9);
		t = now();
		(void)hipMemcpy(dev, pinned, bytes, hipMemcpyHostToDevice);
		printf("H2D pinned:   %.1f ms  %.1f GB/s\n", (now() - t) * 1e3, bytes / (now() - t) / 1e9);
		t = now();
		(void)hipMemcpy(pageable, dev, bytes, hipMemcpyDeviceToHost);
		printf("D2H pageable: %.1f ms  %.1f GB/s\n", (now() - t) * 1e3, bytes / (now() - t) / 1e9);
		t = now();
		(void)hipMemcpy(pinned, dev, bytes, hipMemcpyDeviceToHost);
		printf("D2H pinned:   %.1f ms  %.1f GB/s\n", (now() - t) * 1e3, bytes / (now() - t) / 1e9);
	}
	for (int T : {1, 2, 4, 8, 16, 32}) {
		double best = 1e9;
		for (int rep = 0; rep < 3; ++rep) {
			const double t = now();
			std::vector<std::thread> th;
			for (int i = 0; i < T; ++i)
				th.emplace_back([=] { memcpy(pinned + bytes / T * i, pageable + bytes / T * i, bytes / T); });
			for (auto &x : th)
				x.join();
			best = std::min(best, now() - t);
		}
		printf("memcpy pageable -> pinned, %2d threads: %.1f ms  %.1f GB/s\n", T, best * 1e3, bytes / best / 1e9);
	}
	{
		double t = now();
		(void)hipHostRegister(pageable, bytes, hipHostRegisterDefault);
		printf("hipHostRegister 1 GiB: %.1f ms\n", (now() - t) * 1e3);
		t = now();
		(void)hipMemcpy(dev, pageable, bytes, hipMemcpyHostToDevice);
		printf("H2D registered: %.1f ms  %.1f GB/s\n", (now() - t) * 1e3, bytes / (now() - t) / 1e9);
		t = now();
		(void)hipHostUnregister(pageable);
		printf("hipHostUnregister: %.1f ms\n", (now() - t) * 1e3);
	}
	return 0;
}
