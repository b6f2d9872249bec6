// fmt_g6.h -- a double as C++ streams print it by default (precision 6, "%g"): what every number of the reference's
// plane_refinement_inliers.xyz looks like (wass_stereo.cpp:2077-2085 writes "x y z" per line through a default ofstream).
// Written so that the GPU can produce that file's text itself: 1.4 million numbers per 5-megapixel frame took 37 ms of host
// time per frame in round 4, a third of a worker's CPU.
//
// EXACT: the six significant digits are the correctly rounded (round-half-even on the exact binary value, like glibc's printf
// and std::to_chars) decimal digits, decided in 128-bit integer arithmetic -- a double product a * 10^k is off by half an ulp,
// which is enough to pick the wrong digit next to a tie, and ties are not rare among coordinates that are dyadic rationals
// (1.015625 -> "1.01562").  Domain: 0, and 1e-22 <= |v| < 1e6 (what a camera-frame coordinate can be: triangulate() rejects
// points farther than 200 units); anything else -- inf, nan, huge, tiny -- returns -1 and the caller falls back to the host
// formatter (hostio.hpp fmt_g6 / std::to_chars).  No libm, no division of wide integers: runs unchanged on gfx950 and on the
// host (tests/test_fmt_g6.py compares the host build with printf on adversarial and random values).
#pragma once

#include <stdint.h>

#ifndef WASS_HD
#ifdef __HIPCC__
#define WASS_HD __host__ __device__
#else
#define WASS_HD
#endif
#endif

namespace wass {

typedef unsigned __int128 u128_t;

WASS_HD inline uint64_t pow5_u64(int k)          // 5^k, 0 <= k <= 27
{
    // 5^13 = 1220703125 < 2^31: two table look-ups of small powers and one multiplication instead of a loop
    constexpr uint64_t P[14] = { 1ull, 5ull, 25ull, 125ull, 625ull, 3125ull, 15625ull, 78125ull, 390625ull, 1953125ull, 9765625ull, 48828125ull,
                                 244140625ull, 1220703125ull };
    const int a = k > 13 ? 13 : k;
    uint64_t r = P[a];
    const int b = k - a;                            // 0 .. 14
    if (b > 0) r *= b > 13 ? P[13] * 5ull : P[b];
    return r;
}

// floor and rounding remainder of m * 2^q * 10^k for 0 <= k <= 27, as long as the result is below 2^40 or so:
// D = floor(value), cmp = sign(value - D - 1/2) (-1 below the half, 0 exactly on it, +1 above)
WASS_HD inline void scaled_digits(uint64_t m, int q, int k, uint64_t& D, int& cmp)
{
    const u128_t N = (u128_t)m * pow5_u64(k);                        // < 2^53 * 2^63 = 2^116
    const int s = q + k;                                             // value = N * 2^s
    if (s >= 0) { D = (uint64_t)(N << s); cmp = -1; return; }        // an integer (s is small here: value < 2^40)
    const int sh = -s;
    if (sh >= 127) { D = 0; cmp = -1; return; }                      // (never for values >= 1)
    D = (uint64_t)(N >> sh);
    const u128_t rem = N & ((((u128_t)1) << sh) - 1), half = ((u128_t)1) << (sh - 1);
    cmp = rem > half ? 1 : (rem == half ? 0 : -1);
}

// Writes the characters of v (no terminator) to out, which must have room for 16; returns their number, or -1 outside the domain.
WASS_HD inline int fmt_g6(double v, char* out)
{
    union { double d; uint64_t u; } b;
    b.d = v;
    const bool neg = (b.u >> 63) != 0;
    const int be = (int)((b.u >> 52) & 0x7FF);
    const uint64_t frac = b.u & 0xFFFFFFFFFFFFFull;
    int n = 0;
    if (be == 0 && frac == 0) {                                      // +-0
        if (neg) out[n++] = '-';
        out[n++] = '0';
        return n;
    }
    if (be == 0x7FF || be == 0) return -1;                           // inf, nan, subnormal
    const double a = neg ? -v : v;
    if (!(a >= 1e-22 && a < 1e6)) return -1;
    const uint64_t m = frac | (1ull << 52);
    const int q = be - 1075;                                         // a = m * 2^q
    // decimal exponent: an estimate from the binary one (log10(2) = 0.30103), made exact with the digits themselves
    int e = (int)(((long long)(be - 1023) * 78913LL) >> 18);         // floor((be-1023) * log10 2), exact or one too small in this range
    if (e < -22) e = -22;                                            // (a >= 1e-22: the estimate was the one too small; keeps 5^(5-e) in 64 bits)
    uint64_t D = 0;
    int cmp = -1;
    for (int it = 0; it < 3; ++it) {
        scaled_digits(m, q, 5 - e, D, cmp);                          // a * 10^(5-e): in [1e5, 1e6) when e is right
        if (D >= 1000000u) ++e;
        else if (D < 100000u) --e;
        else break;
    }
    if (D < 100000u || D >= 1000000u) return -1;                     // (cannot happen)
    if (cmp > 0 || (cmp == 0 && (D & 1u))) ++D;                      // round half to even
    if (D == 1000000u) { D = 100000u; ++e; }                         // the rounding carried into the next decade
    char dg[6];
    {
        uint32_t t = (uint32_t)D;
        for (int i = 5; i >= 0; --i) { dg[i] = (char)('0' + t % 10u); t /= 10u; }
    }
    int nd = 6;
    while (nd > 1 && dg[nd - 1] == '0') --nd;                        // %g strips trailing zeros
    if (neg) out[n++] = '-';
    if (e >= -4 && e < 6) {                                          // fixed notation
        if (e >= 0) {
            const int ip = e + 1;                                    // digits in front of the point (zeros among them are kept)
            for (int i = 0; i < ip; ++i) out[n++] = dg[i];
            if (nd > ip) { out[n++] = '.'; for (int i = ip; i < nd; ++i) out[n++] = dg[i]; }
        } else {
            out[n++] = '0'; out[n++] = '.';
            for (int i = 0; i < -e - 1; ++i) out[n++] = '0';
            for (int i = 0; i < nd; ++i) out[n++] = dg[i];
        }
    } else {                                                         // d[.ddddd]e+-XX
        out[n++] = dg[0];
        if (nd > 1) { out[n++] = '.'; for (int i = 1; i < nd; ++i) out[n++] = dg[i]; }
        out[n++] = 'e';
        int x = e;
        if (x < 0) { out[n++] = '-'; x = -x; } else out[n++] = '+';
        out[n++] = (char)('0' + x / 10);                             // |e| <= 22 here: two digits
        out[n++] = (char)('0' + x % 10);
    }
    return n;
}

}  // namespace wass
