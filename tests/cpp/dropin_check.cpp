// dropin_check.cpp -- exercises the C++ template surface in include/ (radix_sort.hpp,
// radix_sort_rank.hpp, radix_sort_basic_kdf.hpp) the way the reference's own callers do
// (radix_tests.cpp:45-207 shapes, radix_experiment.cpp:205), but checks exact results: stable order,
// returned-pointer parity, untouched aux on the early exits.  Needs a GPU; run by tests/test_gpu_cpp.py.
#include <algorithm>
#include <array>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <random>
#include <string>
#include <thread>
#include <vector>

#include "radix_sort.hpp"
#include "radix_sort_rank.hpp"

static int failures = 0;
#define CHECK(cond)                                                         \
	do {                                                                    \
		if (!(cond)) {                                                      \
			printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond);        \
			++failures;                                                     \
		}                                                                   \
	} while (0)

struct sortrec {
	uint8_t key;
	const char *name;
};
static const sortrec source_arr[] = {{255, "1st 255"}, {45, "1st 45"}, {3, "3"},  {45, "2nd 45"},
                                     {2, "2"},         {45, "3rd 45"}, {1, "1"}, {255, "2nd 255"}};

// ---- KeyFuncs that are plain functions (the reference takes any callable, radix_sort.hpp:98-99; the shapes of
// radix_tests.cpp:111-113,:175-177 written as free functions).  Each has the TYPE of the default basic_kdfs::kdf<T>.
static uint32_t fn_u32_desc(const uint32_t &v) { return ~v; }
static uint32_t fn_u32_flip(const uint32_t &v) { return v ^ 0x80000000u; }
static uint32_t fn_u32_ident(const uint32_t &v) { return v; }
static uint32_t fn_i32_desc(const int32_t &v) { return ~((uint32_t)v ^ 0x80000000u); }
static uint32_t fn_i32_plain(const int32_t &v) { return (uint32_t)v; }              // orders negatives after positives
static uint32_t fn_f32_desc(const float &v) { return ~basic_kdfs::kdf(v); }
static uint32_t fn_f32_bits(const float &v) { uint32_t u; std::memcpy(&u, &v, 4); return u; }
static uint64_t fn_u64_desc(const uint64_t &v) { return ~v; }

// radix_sort and radix_sort_rank with `fn` passed as a function, as a function pointer and wrapped in a lambda must all
// give std::stable_sort's order by fn
template <typename T, typename K>
static void check_free_function(K (&fn)(const T &), size_t n, uint64_t mask, unsigned seed)
{
	std::mt19937_64 rng(seed);
	std::vector<T> in(n);
	for (size_t i = 0; i < n; ++i) {
		uint64_t bits = rng() & mask;
		std::memcpy(&in[i], &bits, sizeof(T));
	}
	std::vector<T> want = in;
	std::stable_sort(want.begin(), want.end(), [&](const T &a, const T &b) { return fn(a) < fn(b); });
	std::vector<uint32_t> wr(n);
	std::iota(wr.begin(), wr.end(), 0u);
	std::stable_sort(wr.begin(), wr.end(), [&](uint32_t a, uint32_t b) { return fn(in[a]) < fn(in[b]); });
	{
		std::vector<T> src = in, aux(n);
		T *res = radix_sort(src.data(), aux.data(), n, fn);                  // KeyFunc = K (&)(const T &)
		CHECK(std::memcmp(res, want.data(), n * sizeof(T)) == 0);
	}
	{
		std::vector<T> src = in, aux(n);
		K (*fp)(const T &) = fn;
		T *res = radix_sort(src.data(), aux.data(), n, fp);                  // KeyFunc = K (*&)(const T &)
		CHECK(std::memcmp(res, want.data(), n * sizeof(T)) == 0);
		T *res2 = radix_sort(src.data(), aux.data(), 0, &fn);                // rvalue pointer, n = 0
		CHECK(res2 == src.data());
	}
	{
		std::vector<uint32_t> ib(2 * n);
		uint32_t *ranks = radix_sort_rank(in.data(), ib.data(), n, fn);
		CHECK(std::memcmp(ranks, wr.data(), n * 4) == 0);
		K (*fp)(const T &) = fn;
		std::vector<uint32_t> ib2(2 * n);
		uint32_t *ranks2 = radix_sort_rank(in.data(), ib2.data(), n, fp);
		CHECK(std::memcmp(ranks2, wr.data(), n * 4) == 0);
	}
}

// FNV-1a-64 (as oracle/rs_oracle.c, rso_fnv1a64) and the splitmix64 generator of SURVEY.md section 4
static uint64_t fnv1a64(const void *p, size_t bytes)
{
	uint64_t h = 0xcbf29ce484222325ull;
	for (size_t i = 0; i < bytes; ++i)
		h = (h ^ ((const unsigned char *)p)[i]) * 0x100000001b3ull;
	return h;
}
template <typename T> static std::vector<T> splitmix_fill(size_t n, uint64_t seed, uint64_t mask)
{
	std::vector<T> a(n);
	uint64_t st = seed;
	for (size_t i = 0; i < n; ++i) {
		uint64_t z = (st += 0x9E3779B97F4A7C15ull);
		z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
		z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
		z = (z ^ (z >> 31)) & mask;
		std::memcpy(&a[i], &z, sizeof(T));
	}
	return a;
}

// rs_sort_main / rs_sort_rank with the caller's Hist (radix_sort.hpp:28-33): prints what is left in the histogram storage
// as "HIST <dtype code> <n> <seed> <mask> <presorted> <hvt bytes> <fnv of result> <in_aux> <fnv of the Hist as uint64>";
// tests/test_gpu_cpp.py compares the lines with tests/golden/kat_table.json (hist_post: generated from the real reference).
template <typename T, typename HVT, typename Hist>
static void hist_row(int dtype_code, size_t n, uint64_t seed, uint64_t mask, int presorted, Hist &hist)
{
	std::vector<T> src = splitmix_fill<T>(n, seed, mask), aux(n);
	if (presorted)
		std::stable_sort(src.begin(), src.end(), [](const T &a, const T &b) { return basic_kdfs::kdf(a) < basic_kdfs::kdf(b); });
	const std::vector<T> in = src;
	T *res = rs_sort_main(src.data(), aux.data(), n, hist);
	std::vector<uint64_t> h64(256 * sizeof(T));
	for (size_t i = 0; i < h64.size(); ++i)
		h64[i] = hist[i];
	printf("HIST %d %zu %llu %016llx %d %zu %016llx %d %016llx\n", dtype_code, n, (unsigned long long)seed, (unsigned long long)mask,
	       presorted, sizeof(HVT), (unsigned long long)fnv1a64(res, n * sizeof(T)), res == aux.data() ? 1 : 0,
	       (unsigned long long)fnv1a64(h64.data(), h64.size() * 8));
	// rs_sort_rank leaves the same histogram (radix_sort_rank.hpp:44-88 has the same three loops) ...
	std::vector<HVT> hr(256 * sizeof(T), 0);
	std::vector<size_t> ib(2 * n + 1);
	size_t *ranks = rs_sort_rank(in.data(), ib.data(), n, hr);
	bool same = true;
	for (size_t i = 0; i < h64.size(); ++i)
		same &= (uint64_t)hr[i] == h64[i];
	CHECK(same);
	// ... and its ranks are the stable argsort
	std::vector<size_t> wr(n);
	std::iota(wr.begin(), wr.end(), (size_t)0);
	std::stable_sort(wr.begin(), wr.end(), [&](size_t a, size_t b) { return basic_kdfs::kdf(in[a]) < basic_kdfs::kdf(in[b]); });
	CHECK(n == 0 || std::memcmp(ranks, wr.data(), n * sizeof(size_t)) == 0);
}

template <typename T> static void hist_rows(int dtype_code)
{
	const uint64_t FULL = ~0ull, seed = 40 + dtype_code;   // tools/gen_golden.py, HIST_ROWS
	{
		std::array<uint8_t, 256 * sizeof(T)> h{};           // n < 256: radix_sort.hpp:103-105
		hist_row<T, uint8_t>(dtype_code, 200, seed, FULL, 0, h);
	}
	{
		std::array<uint16_t, 256 * sizeof(T)> h{};
		hist_row<T, uint16_t>(dtype_code, 3000, seed, FULL, 0, h);
	}
	{
		std::vector<uint32_t> h(256 * sizeof(T), 0);
		hist_row<T, uint32_t>(dtype_code, 100000, seed, FULL, 0, h);
	}
	{
		std::array<uint32_t, 256 * sizeof(T)> h{};
		hist_row<T, uint32_t>(dtype_code, 100000, seed, 0x00FFFF00FF00FFFFull, 0, h);
	}
	{
		std::vector<uint16_t> h(256 * sizeof(T), 0);
		hist_row<T, uint16_t>(dtype_code, 5000, seed, FULL, 1, h);
	}
	{
		std::vector<uint16_t> h(256 * sizeof(T), 0);
		hist_row<T, uint16_t>(dtype_code, 300, seed, 0xFF, 0, h);
	}
}

template <typename T>
static void check_scalar(size_t n, uint64_t mask, unsigned seed)
{
	std::mt19937_64 rng(seed);
	std::vector<T> src(n), aux(n, T()), want(n);
	for (size_t i = 0; i < n; ++i) {
		uint64_t bits = rng() & mask;
		std::memcpy(&src[i], &bits, sizeof(T));
	}
	want = src;
	std::stable_sort(want.begin(), want.end(), [](const T &a, const T &b) { return basic_kdfs::kdf(a) < basic_kdfs::kdf(b); });
	// number of non-constant key bytes decides the returned buffer (radix_sort.hpp:64-70,:89-92)
	int cols = 0;
	bool presorted = true;
	for (size_t i = 1; i < n; ++i)
		presorted &= basic_kdfs::kdf(src[i - 1]) <= basic_kdfs::kdf(src[i]);
	for (size_t b = 0; b < sizeof(T); ++b) {
		bool varies = false;
		for (size_t i = 1; i < n && !varies; ++i)
			varies = ((basic_kdfs::kdf(src[i]) >> (8 * b)) & 0xFF) != ((basic_kdfs::kdf(src[0]) >> (8 * b)) & 0xFF);
		cols += varies;
	}
	T *res = radix_sort(src.data(), aux.data(), n);
	CHECK(std::memcmp(res, want.data(), n * sizeof(T)) == 0);
	if (n < 2 || presorted)
		CHECK(res == src.data());
	else
		CHECK(res == ((cols & 1) ? aux.data() : src.data()));
	// descending tag: complemented KDF, equal keys keep input order
	std::vector<T> s2(n), a2(n);
	for (size_t i = 0; i < n; ++i) {
		uint64_t bits = rng() & mask;
		std::memcpy(&s2[i], &bits, sizeof(T));
	}
	std::vector<T> w2 = s2;
	std::stable_sort(w2.begin(), w2.end(), [](const T &a, const T &b) { return basic_kdfs::kdf(a) > basic_kdfs::kdf(b); });
	T *r2 = radix_sort(s2.data(), a2.data(), n, rsx_kdf::descending<T>());
	CHECK(std::memcmp(r2, w2.data(), n * sizeof(T)) == 0);
	// ranks of the (unsorted) second array: stable argsort, returned half by column parity
	if (n < (1u << 20)) {
		std::vector<T> orig(n);
		std::vector<uint32_t> ib(2 * n + 2, 0xEEEEEEEEu), wr(n);
		std::mt19937_64 rng2(seed + 1000);
		for (size_t i = 0; i < n; ++i) {
			uint64_t bits = rng2() & mask;
			std::memcpy(&orig[i], &bits, sizeof(T));
		}
		std::iota(wr.begin(), wr.end(), 0u);
		std::stable_sort(wr.begin(), wr.end(), [&](uint32_t a, uint32_t b) { return basic_kdfs::kdf(orig[a]) < basic_kdfs::kdf(orig[b]); });
		uint32_t *ranks = radix_sort_rank(orig.data(), ib.data(), n);
		CHECK(ranks == ib.data() || ranks == ib.data() + n);
		CHECK(n == 0 || std::memcmp(ranks, wr.data(), n * 4) == 0);
		if (n == 0)
			CHECK(ib[0] == 0xEEEEEEEEu);
		if (n == 1)
			CHECK(ib[0] == 0 && ib[1] == 0xEEEEEEEEu);
	}
}

int main()
{
	// radix_tests.cpp:45-69: records ordered by a 1-byte key (one column -> result in aux), stable
	{
		const size_t N = sizeof(source_arr) / sizeof(source_arr[0]);
		std::vector<sortrec> src(source_arr, source_arr + N), aux(N);
		auto kdf_sortrec = [](const sortrec &e) -> uint8_t { return e.key; };
		sortrec *res = radix_sort(src.data(), aux.data(), N, kdf_sortrec);
		const char *expect[] = {"1", "2", "3", "1st 45", "2nd 45", "3rd 45", "1st 255", "2nd 255"};
		CHECK(res == aux.data());
		for (size_t i = 0; i < N; ++i)
			CHECK(std::strcmp(res[i].name, expect[i]) == 0);
	}
	// radix_tests.cpp:121-146: pointers to records, descending via ~key
	{
		const size_t N = sizeof(source_arr) / sizeof(source_arr[0]);
		std::vector<const sortrec *> src(N), aux(N);
		for (size_t i = 0; i < N; ++i)
			src[i] = &source_arr[i];
		auto kdf_rev = [](const sortrec *e) -> uint8_t { return ~e->key; };
		const sortrec **res = radix_sort(src.data(), aux.data(), N, kdf_rev);
		const char *expect[] = {"1st 255", "2nd 255", "1st 45", "2nd 45", "3rd 45", "3", "2", "1"};
		for (size_t i = 0; i < N; ++i)
			CHECK(std::strcmp(res[i]->name, expect[i]) == 0);
	}
	// radix_tests.cpp:156-173 + README.md:612-623: float order including -0.0, infinities, NaN
	{
		float src[] = {128.0f, 646464.0f, 0.0f, -0.0f, -0.5f, 0.5f, -128.0f, -INFINITY, NAN, INFINITY};
		const size_t N = sizeof(src) / sizeof(src[0]);
		float aux[N];
		float *res = radix_sort(src, aux, N);
		const uint32_t expect[] = {0xff800000, 0xc3000000, 0xbf000000, 0x80000000, 0x00000000,
		                           0x3f000000, 0x43000000, 0x491dd400, 0x7f800000, 0x7fc00000};
		for (size_t i = 0; i < N; ++i) {
			uint32_t u;
			std::memcpy(&u, res + i, 4);
			CHECK(u == expect[i]);
		}
	}
	// radix_tests.cpp:179-207: 50 000 ints ascending, then re-sorted descending with ~(e ^ 1<<31)
	{
		std::default_random_engine generator;
		std::normal_distribution<double> distribution(0.0, 1.0e9);
		const size_t N = 50000;
		std::vector<int> buf(2 * N);
		int *src = buf.data(), *aux = src + N;   // both halves of one allocation (radix_tests.cpp:184-185)
		for (size_t i = 0; i < N; ++i)
			src[i] = (int)std::max(-2.0e9, std::min(2.0e9, distribution(generator)));
		int *res = radix_sort(src, aux, N);
		CHECK(std::is_sorted(res, res + N));
		auto kdf_int_reverse = [](const int &e) -> unsigned int { return ~((unsigned)e ^ (1u << 31)); };
		res = radix_sort(res, res == src ? aux : src, N, kdf_int_reverse);
		CHECK(std::is_sorted(res, res + N, std::greater<int>()));
	}
	// radix_tests.cpp:71-105: rank sort of the records with IdxType = uint8_t
	{
		const size_t N = sizeof(source_arr) / sizeof(source_arr[0]);
		std::vector<uint8_t> ib(2 * N, 0xEE);
		auto kdf_sortrec = [](const sortrec &e) -> uint8_t { return e.key; };
		uint8_t *ranks = radix_sort_rank(source_arr, ib.data(), N, kdf_sortrec);
		const uint8_t expect[] = {6, 4, 2, 1, 3, 5, 0, 7};
		CHECK(ranks == ib.data() + N);
		for (size_t i = 0; i < N; ++i)
			CHECK(ranks[i] == expect[i]);
	}
	// scalars of every width: exact stable order and returned-pointer parity, with skipped columns
	check_scalar<uint32_t>(100003, 0xFFFFFFFFull, 1);
	check_scalar<uint32_t>(100003, 0x00FFFFFFull, 2);
	check_scalar<uint32_t>(70000, 0x0000FF00ull, 3);
	check_scalar<uint64_t>(65537, 0x000000FFFFFFFFFFull, 4);
	check_scalar<int64_t>(50000, ~0ull, 5);
	check_scalar<int32_t>(4097, ~0ull, 6);
	check_scalar<uint16_t>(30000, 0xFFFFull, 7);
	check_scalar<uint8_t>(30000, 0xFFull, 8);
	check_scalar<int8_t>(255, 0xFFull, 9);
	check_scalar<float>(100000, 0xFFFFFFFFull, 10);
	check_scalar<double>(100000, ~0ull, 11);
	check_scalar<uint32_t>(1, ~0ull, 12);
	check_scalar<uint32_t>(0, ~0ull, 13);
	// float ranks against std::stable_sort
	{
		const size_t n = 200000;
		std::mt19937 rng(99);
		std::vector<float> keys(n);
		for (auto &k : keys) {
			uint32_t b = rng() & 0xFFF000FFu;
			std::memcpy(&k, &b, 4);
		}
		std::vector<uint32_t> ib(2 * n), want(n);
		std::iota(want.begin(), want.end(), 0u);
		std::stable_sort(want.begin(), want.end(), [&](uint32_t a, uint32_t b) { return basic_kdfs::kdf(keys[a]) < basic_kdfs::kdf(keys[b]); });
		uint32_t *ranks = radix_sort_rank(keys.data(), ib.data(), n);
		CHECK(std::memcmp(ranks, want.data(), n * 4) == 0);
	}
	// records ordered by a declared member (rsx_kdf::by_member): device extraction + rank + gather, against
	// std::stable_sort; field types, descending, odd record sizes, the returned-pointer rule
	{
		struct Rec16 { uint8_t name[7]; uint8_t pad; float score; uint32_t id; };      // 16 bytes, float key at offset 8
		struct Rec12 { uint16_t tag; int16_t key; uint64_t payload; } __attribute__((packed));   // 12 bytes, unaligned payload
		struct Rec24 { uint64_t a; double key; uint32_t b; uint32_t c; };
		const size_t n = 150001;
		std::mt19937_64 rng(2024);
		{
			std::vector<Rec16> src(n), aux(n), want;
			for (size_t i = 0; i < n; ++i) {
				uint32_t bits = (uint32_t)rng() & 0xFFF000FFu;
				std::memcpy(&src[i].score, &bits, 4);
				src[i].id = (uint32_t)i;
				std::memset(src[i].name, (int)(i & 0x7F), 7);
				src[i].pad = 0;
			}
			want = src;
			std::stable_sort(want.begin(), want.end(), [](const Rec16 &x, const Rec16 &y) { return basic_kdfs::kdf(x.score) < basic_kdfs::kdf(y.score); });
			Rec16 *r = radix_sort(src.data(), aux.data(), n, rsx_kdf::by_member<&Rec16::score>{});
			CHECK(std::memcmp(r, want.data(), n * sizeof(Rec16)) == 0);
			// descending by the same field: equal keys keep forward input order (README.md:564-574)
			std::vector<Rec16> src2(n), aux2(n);
			for (size_t i = 0; i < n; ++i)
				src2[i] = want[n - 1 - i], src2[i].id = (uint32_t)i;
			auto want2 = src2;
			std::stable_sort(want2.begin(), want2.end(), [](const Rec16 &x, const Rec16 &y) { return basic_kdfs::kdf(x.score) > basic_kdfs::kdf(y.score); });
			Rec16 *r2 = radix_sort(src2.data(), aux2.data(), n, rsx_kdf::by_member<&Rec16::score, true>{});
			CHECK(std::memcmp(r2, want2.data(), n * sizeof(Rec16)) == 0);
		}
		{
			std::vector<Rec12> src(n), aux(n), want;
			for (size_t i = 0; i < n; ++i) {
				src[i].tag = (uint16_t)i;
				src[i].key = (int16_t)(rng() & 0xFFFF);
				src[i].payload = rng();
			}
			want = src;
			std::stable_sort(want.begin(), want.end(), [](const Rec12 &x, const Rec12 &y) { return x.key < y.key; });
			Rec12 *r = radix_sort(src.data(), aux.data(), n, rsx_kdf::by_member<&Rec12::key>{});
			CHECK(r == src.data());               // int16: two kept columns -> source buffer
			CHECK(std::memcmp(r, want.data(), n * sizeof(Rec12)) == 0);
		}
		{
			std::vector<Rec24> src(n), aux(n), want;
			for (size_t i = 0; i < n; ++i) {
				src[i].a = rng();
				uint64_t bits = rng() & 0x7FFFFFFFFFFFFF00ull;     // positive, low byte constant: 7 kept columns -> auxiliary buffer
				std::memcpy(&src[i].key, &bits, 8);
				src[i].b = (uint32_t)i;
				src[i].c = ~(uint32_t)i;
			}
			want = src;
			std::stable_sort(want.begin(), want.end(), [](const Rec24 &x, const Rec24 &y) { return basic_kdfs::kdf(x.key) < basic_kdfs::kdf(y.key); });
			Rec24 *r = radix_sort(src.data(), aux.data(), n, rsx_kdf::by_member<&Rec24::key>{});
			CHECK(r == aux.data());
			CHECK(std::memcmp(r, want.data(), n * sizeof(Rec24)) == 0);
			// already sorted by the key: source returned, auxiliary buffer untouched
			std::vector<Rec24> aux2(n);
			std::memset(aux2.data(), 0x5A, n * sizeof(Rec24));
			auto sorted = want;
			Rec24 *r3 = radix_sort(sorted.data(), aux2.data(), n, rsx_kdf::by_member<&Rec24::key>{});
			CHECK(r3 == sorted.data());
			bool untouched = true;
			for (size_t i = 0; i < n * sizeof(Rec24); ++i)
				untouched &= reinterpret_cast<unsigned char *>(aux2.data())[i] == 0x5A;
			CHECK(untouched);
		}
	}
	// The reference is re-entrant: concurrent calls on disjoint buffers are safe.  Six host threads, different types and
	// sizes, several sorts each, all on the default stream's context.
	{
		std::vector<std::thread> th;
		std::vector<int> bad(6, 0);
		for (int t = 0; t < 6; ++t)
			th.emplace_back([t, &bad] {
				std::mt19937_64 rng(1000 + t);
				for (int rep = 0; rep < 6; ++rep) {
					const size_t n = 1000 + (size_t)(rng() % 400000);
					if (t % 3 == 0) {
						std::vector<uint32_t> a(n), b(n);
						for (auto &x : a)
							x = (uint32_t)rng();
						auto want = a;
						std::sort(want.begin(), want.end());
						uint32_t *r = radix_sort(a.data(), b.data(), n);
						bad[t] += std::memcmp(r, want.data(), n * 4) != 0;
					} else if (t % 3 == 1) {
						std::vector<double> a(n), b(n);
						for (auto &x : a)
							x = (double)(int64_t)rng() * 1e-3;
						auto want = a;
						std::sort(want.begin(), want.end());
						double *r = radix_sort(a.data(), b.data(), n);
						bad[t] += std::memcmp(r, want.data(), n * 8) != 0;
					} else {
						std::vector<int16_t> a(n);
						std::vector<uint32_t> ib(2 * n), want(n);
						for (auto &x : a)
							x = (int16_t)rng();
						std::iota(want.begin(), want.end(), 0u);
						std::stable_sort(want.begin(), want.end(), [&](uint32_t x, uint32_t y) { return a[x] < a[y]; });
						uint32_t *r = radix_sort_rank(a.data(), ib.data(), n);
						bad[t] += std::memcmp(r, want.data(), n * 4) != 0;
					}
				}
			});
		for (auto &x : th)
			x.join();
		for (int t = 0; t < 6; ++t)
			CHECK(bad[t] == 0);
	}
	// radix_sort_multi: the same sort spread over three ranks of this process (all on device 0 here); same result, same
	// returned pointer as radix_sort
	{
		const size_t N = 300007;
		std::vector<int32_t> a(N), aux(N), b, baux(N);
		uint64_t s = 99;
		for (auto &x : a) {
			s = s * 6364136223846793005ull + 1442695040888963407ull;
			x = (int32_t)(s >> 33) - (1 << 30);
		}
		b = a;
		const int devs[3] = {0, 0, 0};
		int32_t *r1 = radix_sort_multi(a.data(), aux.data(), N, devs, 3);
		int32_t *r2 = radix_sort(b.data(), baux.data(), N);
		CHECK((r1 == a.data()) == (r2 == b.data()));
		CHECK(std::memcmp(r1, r2, N * sizeof(int32_t)) == 0);
		std::vector<float> f(N), faux(N), g, gaux(N);
		for (size_t i = 0; i < N; ++i)
			f[i] = (float)a[i] * 0.25f;      // a is sorted now: descending order still has to move everything
		g = f;
		float *d1 = radix_sort_multi(f.data(), faux.data(), N, devs, 3, rsx_kdf::descending<float>());
		float *d2 = radix_sort(g.data(), gaux.data(), N, rsx_kdf::descending<float>());
		CHECK(std::memcmp(d1, d2, N * sizeof(float)) == 0);
		CHECK(d1[0] >= d1[N - 1]);
	}
	// a plain function as KeyFunc has the default KDF's type but not its order (radix_sort.hpp:98-99 takes any callable)
	check_free_function<uint32_t>(fn_u32_desc, 100003, 0xFFFFFFFFull, 21);
	check_free_function<uint32_t>(fn_u32_flip, 70001, 0xFFFFFFFFull, 22);
	check_free_function<uint32_t>(fn_u32_ident, 70001, 0x00FFFFFFull, 23);
	check_free_function<uint32_t>(fn_u32_desc, 300, 0xFFFFull, 24);
	check_free_function<int32_t>(fn_i32_desc, 100003, 0xFFFFFFFFull, 25);
	check_free_function<int32_t>(fn_i32_plain, 5000, 0xFFFFFFFFull, 26);
	check_free_function<float>(fn_f32_desc, 100003, 0xFFFFFFFFull, 27);
	check_free_function<float>(fn_f32_bits, 65536, 0xFFF000FFull, 28);
	check_free_function<uint64_t>(fn_u64_desc, 50000, 0xFFFFFFFFFFull, 29);
	{
		// ... while the default itself, named explicitly as a function or through a pointer, stays the default
		std::vector<int32_t> a = splitmix_fill<int32_t>(50001, 5, ~0ull), aux(a.size()), want = a;
		std::sort(want.begin(), want.end());
		int32_t *r = radix_sort(a.data(), aux.data(), a.size(), basic_kdfs::kdf<int32_t>);
		CHECK(std::memcmp(r, want.data(), a.size() * 4) == 0);
		std::vector<int32_t> b = splitmix_fill<int32_t>(50001, 5, ~0ull);
		auto *fp = &basic_kdfs::kdf<int32_t>;
		r = radix_sort(b.data(), aux.data(), b.size(), fp);
		CHECK(std::memcmp(r, want.data(), b.size() * 4) == 0);
	}
	// rs_sort_main / rs_sort_rank with caller-supplied histogram storage (std::array and std::vector; radix_sort.hpp:28-33)
	hist_rows<uint8_t>(0);
	hist_rows<uint16_t>(1);
	hist_rows<uint32_t>(2);
	hist_rows<uint64_t>(3);
	hist_rows<int8_t>(4);
	hist_rows<int16_t>(5);
	hist_rows<int32_t>(6);
	hist_rows<int64_t>(7);
	hist_rows<float>(8);
	hist_rows<double>(9);
	{
		// a lambda KeyFunc with a caller's Hist: one column of a record
		const size_t N = sizeof(source_arr) / sizeof(source_arr[0]);
		std::vector<sortrec> src(source_arr, source_arr + N), aux(N);
		std::array<uint8_t, 256> h{};
		auto kdf_sortrec = [](const sortrec &e) -> uint8_t { return e.key; };
		sortrec *res = rs_sort_main(src.data(), aux.data(), N, h, kdf_sortrec);
		CHECK(res == aux.data());
		CHECK(h[0] == 0 && h[1] == 1 && h[2] == 2 && h[3] == 3 && h[44] == 3 && h[45] == 6 && h[254] == 6 && h[255] == 8);   // end offsets
	}
	{
		// Elements that are movable but not trivially copyable (the reference only move-assigns T, radix_sort.hpp:85-87):
		// records that own a std::string.  Stable order, the reference's returned buffer (src for an even number of kept
		// columns, aux for an odd one, :92), the pre-sorted exit with aux untouched (:60-62).
		struct Owner {
			uint32_t key;
			std::string name;
		};
		for (uint32_t mask : {0xFFFFFFFFu, 0x00FFFFFFu, 0x000000FFu}) {
			const size_t N = 30011;
			std::vector<uint32_t> k = splitmix_fill<uint32_t>(N, 61, mask);
			std::vector<Owner> src(N), aux(N), want(N);
			for (size_t i = 0; i < N; ++i) {
				src[i].key = k[i] & 0xFFFFFF0Fu;      // (duplicates: stability is observable through the names)
				src[i].name = "element #" + std::to_string(i) + " of a sort of records that own their storage";
			}
			want = src;
			std::stable_sort(want.begin(), want.end(), [](const Owner &a, const Owner &b) { return a.key < b.key; });
			auto kf = [](const Owner &e) -> uint32_t { return e.key; };
			Owner *res = radix_sort(src.data(), aux.data(), N, kf);
			const int cols = (mask == 0xFFFFFFFFu) ? 4 : (mask == 0x00FFFFFFu ? 3 : 1);
			CHECK(res == ((cols & 1) ? aux.data() : src.data()));
			bool same = true;
			for (size_t i = 0; i < N; ++i)
				same = same && res[i].key == want[i].key && res[i].name == want[i].name;
			CHECK(same);
			// sorted now: the same buffer comes back, the other one is not touched
			Owner *other = res == src.data() ? aux.data() : src.data();
			for (size_t i = 0; i < N; ++i)
				other[i].name = "untouched";
			Owner *again = radix_sort(res, other, N, kf);
			CHECK(again == res);
			bool untouched = true;
			for (size_t i = 0; i < N; ++i)
				untouched = untouched && other[i].name == "untouched" && res[i].name == want[i].name;
			CHECK(untouched);
		}
	}
	{
		// rs_sort_main's fourth template parameter (radix_sort.hpp:31): an explicit KeyType narrower than what the KDF
		// returns -- the keys are the KDF's values converted to it, its size is the number of columns
		const size_t N = 40001;
		std::vector<uint32_t> src = splitmix_fill<uint32_t>(N, 71, ~0ull), aux(N), want = src;
		std::stable_sort(want.begin(), want.end(), [](uint32_t a, uint32_t b) { return (uint16_t)a < (uint16_t)b; });
		std::vector<uint32_t> h(256 * 2, 0);
		auto kf = [](const uint32_t &v) -> uint32_t { return v; };
		uint32_t *res = rs_sort_main<uint32_t, decltype(kf) &, std::vector<uint32_t>, uint16_t>(src.data(), aux.data(), N, h, kf);
		CHECK(res == src.data());                 // two columns
		CHECK(std::memcmp(res, want.data(), N * 4) == 0);
		CHECK(h[255] == N && h[511] == N);        // end offsets of both columns
	}
	if (failures) {
		printf("dropin_check: %d failures\n", failures);
		return 1;
	}
	printf("dropin_check OK\n");
	return 0;
}
