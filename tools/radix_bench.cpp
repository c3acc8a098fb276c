// radix_bench -- counterpart of the reference's Google-Benchmark program (radix_bench.cpp:43-140)
// without libbenchmark: the same row names (FSu32/<sort>/<n>), the same sizes (1, 10, ..., 10^7, 4*10^7 keys of
// 40M_32bit_keys.dat) and the same columns (Time, CPU, Iterations, KeyRate, bytes_per_second), so that a run can be
// laid next to report/*.txt of the reference.
//
//   ./radix_bench [--ref-loop] [--min-time SECONDS] [--filter SUBSTRING] [--device [N]] [--verify]
//
// Rows: radix_sort and radix_sort_rank go through this repo's include/ headers (MI355X); StdSort and QSort are the
// host's std::sort / qsort, as in the reference.
// Default: every iteration sorts a fresh copy of the input; the copy is outside the timed region.  --ref-loop
// reproduces the reference's loop (radix_bench.cpp:91-93), which sorts the same buffers again and again, i.e. mostly
// the pre-sorted early exit after the first iteration.  --device adds rows that keep the keys in HBM
// (rsx_sort_device on a device copy; no PCIe in the timed region); with a number it also selects the HIP device.
// --verify checks, outside the timed region, the output of every radix row against std::sort of the same keys (ranks:
// against the stable argsort) -- what radix_experiment.cpp:137-174 does for the reference's `radix` -- and prints one
// "verified" line per row; a mismatch ends the run with status 6.
#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include <hip/hip_runtime_api.h>

#include "radix_sort.hpp"
#include "radix_sort_rank.hpp"

typedef std::chrono::steady_clock clk;

static uint64_t splitmix64(uint64_t &s)
{
	uint64_t z = (s += 0x9E3779B97F4A7C15ull);
	z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
	z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
	return z ^ (z >> 31);
}

static std::vector<uint32_t> load_keys(size_t max_n)
{
	std::vector<uint32_t> k(max_n);
	FILE *f = fopen("40M_32bit_keys.dat", "rb");
	if (f) {
		const size_t got = fread(k.data(), 4, max_n, f);
		fclose(f);
		if (got == max_n)
			return k;
	}
	fprintf(stderr, "40M_32bit_keys.dat not found (or short): generating its bytes from splitmix64(seed 40)\n");
	uint64_t s = 40;
	for (size_t i = 0; i + 1 < max_n; i += 2) {
		const uint64_t r = splitmix64(s);
		memcpy(&k[i], &r, 8);
	}
	return k;
}

static double cpu_seconds()
{
	struct timespec ts;
	clock_gettime(CLOCK_PROCESS_CPUTIME_ID, &ts);
	return ts.tv_sec + ts.tv_nsec * 1e-9;
}

static const char *unit(double ns, double *out)
{
	*out = ns;
	return "ns";
}

static void rate(double v, char *buf, size_t len, const char *suffix)
{
	const char *pre[] = {"", "k", "M", "G", "T"};
	int p = 0;
	while (v >= 1000.0 && p < 4) {
		v /= 1000.0;
		++p;
	}
	snprintf(buf, len, "%.4g%s%s", v, pre[p], suffix);
}

struct Opts {
	bool ref_loop = false, device = false, verify = false;
	int device_index = -1;
	double min_time = 0.5;
	std::string filter;
};

static bool selected(const Opts &o, const char *name, size_t n)
{
	char full[96];
	snprintf(full, sizeof full, "FSu32/%s/%zu", name, n);
	return o.filter.empty() || std::string(full).find(o.filter) != std::string::npos;
}

static void verdict(const char *name, size_t n, bool ok, const char *what)
{
	printf("  verified: FSu32/%s/%zu %s %s\n", name, n, ok ? "==" : "DIFFERS FROM", what);
	fflush(stdout);
	if (!ok)
		exit(6);
}

template <typename Prep, typename Body>
static void row(const Opts &o, const char *name, size_t n, size_t bytes_per_key, Prep prep, Body body)
{
	char full[96];
	snprintf(full, sizeof full, "FSu32/%s/%zu", name, n);
	if (!o.filter.empty() && std::string(full).find(o.filter) == std::string::npos)
		return;
	double wall = 0, cpu = 0;
	size_t iters = 0;
	prep();
	body();   // warm-up (context, first-touch)
	if (o.ref_loop)
		prep();
	while ((wall < o.min_time || iters < 3) && iters < 1000000000) {
		if (!o.ref_loop)
			prep();
		const double c0 = cpu_seconds();
		const auto t0 = clk::now();
		body();
		const auto t1 = clk::now();
		cpu += cpu_seconds() - c0;
		wall += std::chrono::duration<double>(t1 - t0).count();
		++iters;
	}
	double t, c;
	const char *u = unit(wall / iters * 1e9, &t);
	unit(cpu / iters * 1e9, &c);
	char kr[32], br[32];
	rate((double)n * iters / wall, kr, sizeof kr, "/s");
	rate((double)n * bytes_per_key * iters / wall, br, sizeof br, "B/s");
	printf("%-34s %12.0f %s %12.0f %s %10zu %12s %14s\n", full, t, u, c, u, iters, kr, br);
	fflush(stdout);
}

int main(int argc, char **argv)
{
	Opts o;
	for (int i = 1; i < argc; ++i) {
		if (!strcmp(argv[i], "--ref-loop"))
			o.ref_loop = true;
		else if (!strcmp(argv[i], "--device")) {
			o.device = true;
			if (i + 1 < argc && argv[i + 1][0] >= '0' && argv[i + 1][0] <= '9')
				o.device_index = atoi(argv[++i]);
		} else if (!strcmp(argv[i], "--verify"))
			o.verify = true;
		else if (!strcmp(argv[i], "--min-time") && i + 1 < argc)
			o.min_time = atof(argv[++i]);
		else if (!strcmp(argv[i], "--filter") && i + 1 < argc)
			o.filter = argv[++i];
		else {
			printf("Usage: %s [--ref-loop] [--min-time SECONDS] [--filter SUBSTRING] [--device [N]] [--verify]\n", argv[0]);
			return 0;
		}
	}
	if (rsx_device_count() <= 0) {
		fprintf(stderr, "no usable gfx950 device: %s\n", rsx_last_error());
		return 3;
	}
	if (o.device_index >= 0 && hipSetDevice(o.device_index) != hipSuccess) {
		fprintf(stderr, "--device %d: no such HIP device\n", o.device_index);
		return 3;
	}
	const size_t max_n = 40000000;
	const std::vector<uint32_t> org = load_keys(max_n);
	std::vector<uint32_t> src(max_n), aux(max_n), idx(2 * max_n);
	std::vector<size_t> sizes;
	for (size_t n = 1; n <= 10000000; n *= 10)
		sizes.push_back(n);
	sizes.push_back(max_n);

	printf("%s\n", o.ref_loop ? "mode: --ref-loop (the same buffers are sorted again and again, as radix_bench.cpp:91-93)"
	                          : "mode: a fresh copy of the input per iteration (copied outside the timed region)");
	printf("%-34s %15s %15s %10s %12s %14s\n", "Benchmark", "Time", "CPU", "Iterations", "KeyRate", "bytes_per_second");
	printf("---------------------------------------------------------------------------------------------------------\n");
	// --verify: std::sort of the first n keys, per size (outside every timed region)
	std::vector<uint32_t> want;
	auto sorted_ref = [&](size_t n) {
		want.assign(org.begin(), org.begin() + n);
		std::sort(want.begin(), want.end());
	};
	for (size_t n : sizes) {
		row(o, "radix_sort", n, 4, [&] { memcpy(src.data(), org.data(), n * 4); },
		    [&] {
			    uint32_t *r = radix_sort(src.data(), aux.data(), n);
			    (void)r;
		    });
		if (o.verify && selected(o, "radix_sort", n)) {
			memcpy(src.data(), org.data(), n * 4);
			const uint32_t *r = radix_sort(src.data(), aux.data(), n);
			sorted_ref(n);
			verdict("radix_sort", n, memcmp(r, want.data(), n * 4) == 0, "std::sort of the same keys");
		}
	}
	for (size_t n : sizes)
		row(o, "StdSort", n, 4, [&] { memcpy(src.data(), org.data(), n * 4); }, [&] { std::sort(src.begin(), src.begin() + n); });
	for (size_t n : sizes)
		row(o, "QSort", n, 4, [&] { memcpy(src.data(), org.data(), n * 4); },
		    [&] {
			    qsort(src.data(), n, 4, [](const void *a, const void *b) {
				    const uint32_t x = *(const uint32_t *)a, y = *(const uint32_t *)b;
				    return x < y ? -1 : x > y;
			    });
		    });
	for (size_t n : sizes) {
		row(o, "radix_sort_rank", n, 4, [&] { memcpy(src.data(), org.data(), n * 4); },
		    [&] {
			    uint32_t *r = radix_sort_rank(src.data(), idx.data(), n);
			    (void)r;
		    });
		if (o.verify && selected(o, "radix_sort_rank", n)) {
			// the stable argsort: keys non-decreasing along the ranks, equal keys by increasing index, every index once
			memcpy(src.data(), org.data(), n * 4);
			const uint32_t *r = radix_sort_rank(src.data(), idx.data(), n);
			bool ok = true;
			for (size_t i = 0; i + 1 < n && ok; ++i)
				ok = r[i] < n && (src[r[i]] < src[r[i + 1]] || (src[r[i]] == src[r[i + 1]] && r[i] < r[i + 1]));
			std::vector<bool> seen(n, false);
			for (size_t i = 0; i < n && ok; ++i) {
				ok = r[i] < n && !seen[r[i]];
				if (ok)
					seen[r[i]] = true;
			}
			verdict("radix_sort_rank", n, ok, "the stable argsort");
		}
	}
	if (o.device) {
		// the same sort with the keys resident in HBM: what a caller gets who keeps its data on the device
		uint32_t *d_org = nullptr, *d_src = nullptr, *d_aux = nullptr;
		if (hipMalloc((void **)&d_org, max_n * 4) != hipSuccess || hipMalloc((void **)&d_src, max_n * 4) != hipSuccess ||
		    hipMalloc((void **)&d_aux, max_n * 4) != hipSuccess ||
		    hipMemcpy(d_org, org.data(), max_n * 4, hipMemcpyHostToDevice) != hipSuccess) {
			fprintf(stderr, "device allocation failed\n");
			return 4;
		}
		for (size_t n : sizes) {
			void *res = nullptr;
			row(o, "radix_sort_device", n, 4,
			    [&] {
				    (void)hipMemcpy(d_src, d_org, n * 4, hipMemcpyDeviceToDevice);
				    (void)hipDeviceSynchronize();
			    },
			    [&] {
				    if (rsx_sort_device(d_src, d_aux, n, RSX_U32, RSX_ASCENDING, nullptr, &res, nullptr) != RSX_OK) {
					    fprintf(stderr, "rsx_sort_device: %s\n", rsx_last_error());
					    exit(5);
				    }
				    (void)hipDeviceSynchronize();
			    });
			if (o.verify && selected(o, "radix_sort_device", n)) {
				(void)hipMemcpy(d_src, d_org, n * 4, hipMemcpyDeviceToDevice);
				if (rsx_sort_device(d_src, d_aux, n, RSX_U32, RSX_ASCENDING, nullptr, &res, nullptr) != RSX_OK ||
				    hipMemcpy(src.data(), res, n * 4, hipMemcpyDeviceToHost) != hipSuccess) {
					fprintf(stderr, "verification sort failed: %s\n", rsx_last_error());
					return 5;
				}
				sorted_ref(n);
				verdict("radix_sort_device", n, memcmp(src.data(), want.data(), n * 4) == 0, "std::sort of the same keys");
			}
		}
		(void)hipFree(d_org);
		(void)hipFree(d_src);
		(void)hipFree(d_aux);
	}
	return 0;
}
