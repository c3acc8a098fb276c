// radix -- counterpart of the reference's `radix` command (radix_experiment.cpp:241-285) on
// top of this repo's include/radix_sort.hpp (MI355X through librsx.so).
//
//   ./radix <count> [<use_mmap> <use_huge> <uint8_t|uint16_t|uint32_t|uint64_t|int32_t|int64_t|float|double> <hex-mask>] [--device N]
//
// Same positional arguments and the same output lines, so that a run can be laid next to a report of the
// reference: the header line (:259), "Allocating ... bytes for ..." (:59), "Applying value mask to input." (:190),
// "Sorting N entries..." (:203), "Verifying sort... Forward sorted OK." (:140-161; always on here), ten head and
// ten tail lines around "[...]" (:107-121,:214-223), "Sorted N entries in X ms" (:228).  <count> = 0 takes the whole
// file; the element count is file bytes / sizeof(T) (:179-186).
// Differences: buffers are plain host allocations (use_mmap / use_huge are accepted and echoed, hugepages do not
// matter to a sort that runs in HBM); if 40M_32bit_keys.dat is not in the current directory its 160 000 000 bytes are
// generated from splitmix64(seed 40) (SURVEY.md 8d cfg 1) instead of failing; two extra lines report the device time.
// The timed region is the same call as in the reference, radix_sort(src, aux, n) on host pointers: it includes the
// PCIe transfers both ways.
#include <algorithm>
#include <chrono>
#include <cinttypes>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include <hip/hip_runtime_api.h>

#include "radix_sort.hpp"

static const char *KEY_FILE = "40M_32bit_keys.dat";
static const size_t KEY_FILE_BYTES = 160000000;

static uint64_t splitmix64(uint64_t &s)
{
	uint64_t z = (s += 0x9E3779B97F4A7C15ull);
	z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
	z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
	return z ^ (z >> 31);
}

// the file's bytes, or as many as `want` (0 = all)
static void *load_keys(size_t *bytes)
{
	FILE *f = fopen(KEY_FILE, "rb");
	size_t have = KEY_FILE_BYTES;
	if (f) {
		fseek(f, 0, SEEK_END);
		have = (size_t)ftell(f);
		fseek(f, 0, SEEK_SET);
	}
	if (*bytes == 0 || *bytes > have)
		*bytes = have;
	printf("Allocating %zu bytes for %s.\n", *bytes, "source buffer");
	void *p = malloc(*bytes ? *bytes : 1);
	if (!p)
		return nullptr;
	if (f) {
		if (fread(p, 1, *bytes, f) != *bytes) {
			fclose(f);
			free(p);
			return nullptr;
		}
		fclose(f);
	} else {
		printf("'%s' not found: generating its bytes from splitmix64(seed 40).\n", KEY_FILE);
		uint64_t s = 40;
		unsigned char *b = (unsigned char *)p;
		size_t i = 0;
		for (; i + 8 <= *bytes; i += 8) {
			const uint64_t r = splitmix64(s);
			memcpy(b + i, &r, 8);
		}
		if (i < *bytes) {
			const uint64_t r = splitmix64(s);
			memcpy(b + i, &r, *bytes - i);
		}
	}
	return p;
}

template <typename T> static void print_range(const T *a, size_t from, size_t count)
{
	for (size_t i = from; i < from + count; ++i) {
		uint64_t bits = 0;
		memcpy(&bits, a + i, sizeof(T));
		printf("%08zu: %08" PRIx64 "\n", i, bits);
	}
}

template <typename T> static int verify(const T *a, size_t n)
{
	printf("Verifying sort... ");
	size_t fwd = 0, rev = 0;
	for (size_t i = 1; i < n; ++i) {
		const auto x = basic_kdfs::kdf<T>(a[i - 1]), y = basic_kdfs::kdf<T>(a[i]);
		if (!fwd && x > y)
			fwd = i;
		if (!rev && x < y)
			rev = i;
	}
	if (fwd == 0) {
		printf("Forward sorted OK.\n");
		return 0;
	}
	if (rev == 0) {
		printf("Reverse sorted OK.\n");
		return 0;
	}
	printf("Forward sort of array invalid at index %zu.\n", fwd);
	return 1;
}

template <typename T> static int run(size_t entries, uint64_t mask)
{
	size_t bytes = sizeof(T) * entries;
	T *src = (T *)load_keys(&bytes);
	if (!src) {
		printf("Error: could not read the keys.\n");
		return 2;
	}
	printf("Allocating %zu bytes for %s.\n", bytes, "auxilary buffer");
	T *aux = (T *)malloc(bytes ? bytes : 1);
	const size_t n = bytes / sizeof(T);
	if (mask != ~0ull) {
		printf("Applying value mask to input.\n");
		for (size_t i = 0; i < n; ++i) {
			uint64_t b = 0;
			memcpy(&b, src + i, sizeof(T));
			b &= mask;
			memcpy(src + i, &b, sizeof(T));
		}
	}
	printf("Sorting %zu entries...\n", n);
	rsx_profile_begin();
	const auto t0 = std::chrono::steady_clock::now();
	T *sorted = radix_sort(src, aux, n);
	const auto t1 = std::chrono::steady_clock::now();
	rsx_profile prof;
	memset(&prof, 0, sizeof prof);
	rsx_profile_end(&prof);
	if (verify(sorted, n) != 0)
		return 1;
	const size_t nprint = 20;
	if (n <= nprint) {
		print_range(sorted, 0, n);
	} else {
		print_range(sorted, 0, nprint / 2);
		printf("[...]\n");
		print_range(sorted, n - nprint / 2, nprint / 2);
	}
	const double ms = std::chrono::duration<double, std::milli>(t1 - t0).count();
	printf("Sorted %zu entries in %.4f ms\n", n, ms);
	printf("Result in the %s buffer; kernels on the device: histogram %.4f ms, %lu scatter pass(es) %.4f ms.\n",
	       sorted == src ? "source" : "auxilary", prof.hist_ms, (unsigned long)prof.scatter_launches, prof.scatter_ms);
	free(src);
	free(aux);
	return 0;
}

int main(int argc0, char *argv0[])
{
	// --device N anywhere on the line selects the HIP device (not in the reference: it has no devices); the remaining
	// arguments are the reference's positional ones (radix_experiment.cpp:241-257)
	int argc = 0, device = -1;
	char *argv[8] = {nullptr};
	for (int i = 0; i < argc0; ++i) {
		if (!strcmp(argv0[i], "--device") && i + 1 < argc0)
			device = atoi(argv0[++i]);
		else if (argc < 8)
			argv[argc++] = argv0[i];
	}
	if (device >= 0 && hipSetDevice(device) != hipSuccess) {
		printf("Error: --device %d: no such HIP device.\n", device);
		return 3;
	}
	const long entries = argc > 1 ? atol(argv[1]) : 0;
	const int use_mmap = argc > 2 ? atoi(argv[2]) : 0;
	const int use_huge = argc > 3 ? atoi(argv[3]) : 0;
	const char *ktype = argc > 4 ? argv[4] : "uint32_t";
	const uint64_t mask = argc > 5 ? strtoull(argv[5], nullptr, 16) : ~0ull;
	if (argc == 1) {
		printf("Usage: %s <count> [<use_mmap> <use_huge> <uint8_t|uint16_t|uint32_t|uint64_t|int32_t|int64_t|float|double> <hex-mask>] [--device N]\n",
		       argv[0]);
		return 0;
	}
	printf("src='%s', entries=%ld, use_mmap=%d, use_huge=%d, type='%s', mask=0x%08lx \n", KEY_FILE, entries, use_mmap, use_huge, ktype,
	       (unsigned long)mask);
	if (rsx_device_count() <= 0) {
		printf("Error: no usable gfx950 device (%s).\n", rsx_last_error());
		return 3;
	}
	const size_t e = entries > 0 ? (size_t)entries : 0;
	static const struct {
		const char *name;
		int (*sort)(size_t, uint64_t);
	} types[] = {{"uint8_t", run<uint8_t>}, {"uint16_t", run<uint16_t>}, {"uint32_t", run<uint32_t>}, {"uint64_t", run<uint64_t>},
	             {"int32_t", run<int32_t>},  {"int64_t", run<int64_t>},   {"float", run<float>},       {"double", run<double>}};
	for (const auto &t : types) {
		if (strcmp(ktype, t.name) != 0)
			continue;
		try {
			return t.sort(e, mask);
		} catch (const std::exception &ex) {
			printf("Error: %s\n", ex.what());
			return 4;
		}
	}
	printf("Error: unknown key type, '%s'.\n", ktype);
	return 100;
}
