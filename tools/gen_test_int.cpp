// Emits the 50 000-int input that the reference's test_int builds
// (radix_tests.cpp:180-190: libstdc++ default_random_engine feeding
// normal_distribution<double>(INT_MIN, INT_MAX), each sample converted to int)
// as raw little-endian int32 on stdout.  The conversion of out-of-range doubles
// is implementation-specific, so the captured bytes are committed as a fixture
// (tests/golden/test_int_input.bin) instead of being regenerated on other hosts.
#include <cstdio>
#include <limits>
#include <random>
#include <vector>

int main()
{
	std::default_random_engine generator;
	std::normal_distribution<double> distribution(std::numeric_limits<int>::min(), std::numeric_limits<int>::max());
	const size_t N = 50000;
	std::vector<int> v(N);
	for (size_t i = 0; i < N; ++i) {
		double a = distribution(generator);
		v[i] = int(a);
	}
	return fwrite(v.data(), sizeof(int), N, stdout) == N ? 0 : 1;
}
