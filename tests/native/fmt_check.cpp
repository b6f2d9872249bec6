// fmt_check.cpp -- test helper: wasshost::fmt_g6 against snprintf("%g") and std::to_chars on many values.
//   fmt_check <count> <seed>   prints the number of mismatches (and the first few) and the share of fast-path results
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>

#include "../../wass_amd/host/hostio.hpp"

int main(int argc, char** argv)
{
    const long n = argc > 1 ? atol(argv[1]) : 1000000;
    std::mt19937_64 rng(argc > 2 ? atol(argv[2]) : 1);
    long bad = 0, shown = 0;
    auto check = [&](double v) {
        char a[64], b[64];
        char* q = wasshost::fmt_g6(a, a + sizeof a - 1, v);
        *q = 0;
        snprintf(b, sizeof b, "%g", v);
        if (strcmp(a, b) != 0) { ++bad; if (shown++ < 10) printf("MISMATCH %.17g: got '%s' want '%s'\n", v, a, b); }
    };
    std::uniform_real_distribution<double> mant(1.0, 10.0), u01(0.0, 1.0);
    for (long i = 0; i < n; ++i) {
        const int e = (int)(rng() % 16) - 7;                              // 1e-7 .. 1e8
        double v = mant(rng) * std::pow(10.0, e);
        if (rng() & 1) v = -v;
        check(v);
        // short decimals, values on and next to rounding boundaries (xxxxx.5 at the sixth digit), exact integers
        const long k = (long)(rng() % 2000000);
        const int sh = (int)(rng() % 10);
        const double t = ((double)k + 0.5) / std::pow(10.0, sh);
        check(t); check(std::nextafter(t, 0.0)); check(std::nextafter(t, 1e300)); check(-t);
        check((double)k / std::pow(10.0, sh));
        check((double)(rng() % 1000001));
        check(std::ldexp((double)(rng() % (1u << 20)), -(int)(rng() % 30)));
    }
    const double specials[] = { 0.0, -0.0, 1e-4, 9.99995e-5, 0.000099999949, 999999.0, 999999.4, 999999.5, 999999.6, 1e6, 123456.5, 100000.0, 99999.95,
                                0.1, 0.3, 1.0, -1.0, 2.5, 1e-300, 1e300, 5e-324, 12345.65, 1234.565, 0.00012345650000000001 };
    for (double v : specials) check(v);
    printf("%ld mismatches\n", bad);
    return bad ? 1 : 0;
}
