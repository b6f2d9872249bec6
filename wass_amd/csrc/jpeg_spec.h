// jpeg_spec.h -- the arithmetic of the debug pictures' JPEG encoder, shared by the host writer (wass_amd/host/jpeg.hpp) and the
// device encoder (jpeg.hip) so that both write the same bytes.  Integer only; every function is its own definition.
//   rgb -> YCbCr      JFIF equations in 16-bit fixed point (0.299 = 19595 / 65536 ...), Cb / Cr offset 128, rounded
//   forward DCT       8 x 8, separable, Loeffler-Ligtenberg-Moshovitz with 13-bit constants, two extra bits after the row pass;
//                     the result is the DCT-II coefficient times 8
//   quantisation      (|c| + 4 q) / (8 q), sign restored: rounding to nearest, halves away from zero
// WASS_JPEG_FN is `inline` on the host and `__device__ __forceinline__` on the GPU.
#pragma once

#include <stdint.h>

#include <vector>

#ifndef WASS_JPEG_FN
#define WASS_JPEG_FN inline
#endif

namespace wassjpeg {

static const uint8_t kZigzag[64] = { 0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
                                     35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63 };
// Annex K.1 / K.2 quantisation tables (natural order)
static const uint8_t kQLum[64] = { 16, 11, 10, 16, 24, 40, 51, 61, 12, 12, 14, 19, 26, 58, 60, 55, 14, 13, 16, 24, 40, 57, 69, 56, 14, 17, 22, 29, 51, 87, 80, 62,
                                   18, 22, 37, 56, 68, 109, 103, 77, 24, 35, 55, 64, 81, 104, 113, 92, 49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99 };
static const uint8_t kQChr[64] = { 17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99, 24, 26, 56, 99, 99, 99, 99, 99, 47, 66, 99, 99, 99, 99, 99, 99,
                                   99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99 };
// Annex K.3: number of codes of each length 1..16, then the symbols in code order
static const uint8_t kDcLumBits[16] = { 0, 1, 5, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0 };
static const uint8_t kDcChrBits[16] = { 0, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0 };
static const uint8_t kDcVals[12] = { 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11 };
static const uint8_t kAcLumBits[16] = { 0, 2, 1, 3, 3, 2, 4, 3, 5, 5, 4, 4, 0, 0, 1, 0x7d };
static const uint8_t kAcLumVals[162] = {
    0x01, 0x02, 0x03, 0x00, 0x04, 0x11, 0x05, 0x12, 0x21, 0x31, 0x41, 0x06, 0x13, 0x51, 0x61, 0x07, 0x22, 0x71, 0x14, 0x32, 0x81, 0x91, 0xa1, 0x08, 0x23, 0x42, 0xb1,
    0xc1, 0x15, 0x52, 0xd1, 0xf0, 0x24, 0x33, 0x62, 0x72, 0x82, 0x09, 0x0a, 0x16, 0x17, 0x18, 0x19, 0x1a, 0x25, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x34, 0x35, 0x36, 0x37,
    0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a,
    0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3,
    0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3,
    0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe1, 0xe2, 0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf1, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa };
static const uint8_t kAcChrBits[16] = { 0, 2, 1, 2, 4, 4, 3, 4, 7, 5, 4, 4, 0, 1, 2, 0x77 };
static const uint8_t kAcChrVals[162] = {
    0x00, 0x01, 0x02, 0x03, 0x11, 0x04, 0x05, 0x21, 0x31, 0x06, 0x12, 0x41, 0x51, 0x07, 0x61, 0x71, 0x13, 0x22, 0x32, 0x81, 0x08, 0x14, 0x42, 0x91, 0xa1, 0xb1, 0xc1,
    0x09, 0x23, 0x33, 0x52, 0xf0, 0x15, 0x62, 0x72, 0xd1, 0x0a, 0x16, 0x24, 0x34, 0xe1, 0x25, 0xf1, 0x17, 0x18, 0x19, 0x1a, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x35, 0x36,
    0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69,
    0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x82, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a,
    0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca,
    0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe2, 0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa };

struct Huff { uint16_t code[256]; uint8_t len[256]; };
inline Huff make_huff(const uint8_t bits[16], const uint8_t* vals)
{
    Huff h{};
    int code = 0, k = 0;
    for (int l = 1; l <= 16; ++l) {                       // canonical codes: Annex C
        for (int i = 0; i < bits[l - 1]; ++i) { h.code[vals[k]] = (uint16_t)code; h.len[vals[k]] = (uint8_t)l; ++code; ++k; }
        code <<= 1;
    }
    return h;
}
// the four tables every picture uses, in the order [DC luma, DC chroma, AC luma, AC chroma]
struct HuffSet { Huff t[4]; };
inline HuffSet make_huff_set()
{
    HuffSet s;
    s.t[0] = make_huff(kDcLumBits, kDcVals); s.t[1] = make_huff(kDcChrBits, kDcVals);
    s.t[2] = make_huff(kAcLumBits, kAcLumVals); s.t[3] = make_huff(kAcChrBits, kAcChrVals);
    return s;
}

// component c (0 Y, 1 Cb, 2 Cr) of an r,g,b pixel
WASS_JPEG_FN int ycc(int r, int g, int b, int c)
{
    if (c == 0) return (19595 * r + 38470 * g + 7471 * b + 32768) >> 16;
    if (c == 1) return (-11059 * r - 21709 * g + 32768 * b + (128 << 16) + 32767) >> 16;
    return (32768 * r - 27439 * g - 5329 * b + (128 << 16) + 32767) >> 16;
}

WASS_JPEG_FN int descale(int x, int n) { return (x + (1 << (n - 1))) >> n; }

// one 1-D pass over d[0], d[s], ..., d[7 s]; first = the row pass (results carry 2 extra bits), else the column pass (they are removed)
WASS_JPEG_FN void fdct_1d(int* d, int s, bool first)
{
    const int CB = 13, P1 = 2;
    const int t0 = d[0] + d[7 * s], t7 = d[0] - d[7 * s], t1 = d[s] + d[6 * s], t6 = d[s] - d[6 * s];
    const int t2 = d[2 * s] + d[5 * s], t5 = d[2 * s] - d[5 * s], t3 = d[3 * s] + d[4 * s], t4 = d[3 * s] - d[4 * s];
    const int t10 = t0 + t3, t13 = t0 - t3, t11 = t1 + t2, t12 = t1 - t2;
    const int sh = first ? CB - P1 : CB + P1;
    d[0] = first ? (t10 + t11) << P1 : descale(t10 + t11, P1);
    d[4 * s] = first ? (t10 - t11) << P1 : descale(t10 - t11, P1);
    int z1 = (t12 + t13) * 4433;
    d[2 * s] = descale(z1 + t13 * 6270, sh);
    d[6 * s] = descale(z1 - t12 * 15137, sh);
    z1 = t4 + t7;
    int z2 = t5 + t6, z3 = t4 + t6, z4 = t5 + t7;
    const int z5 = (z3 + z4) * 9633;
    const int a4 = t4 * 2446, a5 = t5 * 16819, a6 = t6 * 25172, a7 = t7 * 12299;
    z1 *= -7373; z2 *= -20995; z3 = z3 * -16069 + z5; z4 = z4 * -3196 + z5;
    d[7 * s] = descale(a4 + z1 + z3, sh);
    d[5 * s] = descale(a5 + z2 + z4, sh);
    d[3 * s] = descale(a6 + z2 + z3, sh);
    d[s] = descale(a7 + z1 + z4, sh);
}

// blk: 64 level-shifted samples (value - 128), row-major; on return 8 x the DCT coefficients, row-major (natural order)
WASS_JPEG_FN void fdct8x8(int* blk)
{
    for (int r = 0; r < 8; ++r) fdct_1d(blk + 8 * r, 1, true);
    for (int c = 0; c < 8; ++c) fdct_1d(blk + c, 8, false);
}

// coef = 8 x the DCT coefficient, q = the table entry (1..255); AC amplitudes are clamped to the 10 bits baseline coding allows
WASS_JPEG_FN int quantise(int coef, int q, bool dc)
{
    const int qv = q << 3, a = coef < 0 ? -coef : coef, n = a + (qv >> 1);
#ifdef __HIP_DEVICE_COMPILE__
    // n / qv without the integer-division sequence (64 of them per block): a float estimate corrected by one -- exact for n < 2^24
    int v = (int)((float)n * __frcp_rn((float)qv));
    const int r = n - v * qv;
    v += (r >= qv) - (r < 0);
#else
    int v = n / qv;
#endif
    if (!dc && v > 1023) v = 1023;
    return coef < 0 ? -v : v;
}

// libjpeg's quality scaling of an Annex K table entry
WASS_JPEG_FN int scaled_q(int base, int quality)
{
    const int scale = quality < 50 ? 5000 / (quality < 1 ? 1 : quality) : 200 - 2 * (quality > 100 ? 100 : quality);
    const int v = (base * scale + 50) / 100;
    return v < 1 ? 1 : (v > 255 ? 255 : v);
}

WASS_JPEG_FN int bit_size(int v)
{
    int a = v < 0 ? -v : v;
#ifdef __HIP_DEVICE_COMPILE__
    return 32 - __clz(a);
#else
    int n = 0;
    while (a) { ++n; a >>= 1; }
    return n;
#endif
}

// q[t][i], t = 0 luma / 1 chroma, natural order
inline void quant_tables(int quality, uint8_t q[2][64])
{
    for (int i = 0; i < 64; ++i) { q[0][i] = (uint8_t)scaled_q(kQLum[i], quality); q[1][i] = (uint8_t)scaled_q(kQChr[i], quality); }
}

// Everything in front of the entropy-coded data: SOI, JFIF APP0, DQT, SOF0 (8 bit, no subsampling), DHT, DRI, SOS.
// restart_interval = MCUs per restart interval: one row of blocks, (w + 7) / 8.
inline void file_header(std::vector<uint8_t>& o, int w, int h, int channels, int quality)
{
    uint8_t q[2][64];
    quant_tables(quality, q);
    auto be16 = [&](int v) { o.push_back((uint8_t)(v >> 8)); o.push_back((uint8_t)v); };
    auto marker = [&](uint8_t m) { o.push_back(0xFF); o.push_back(m); };
    marker(0xD8);                                                             // SOI
    marker(0xE0); be16(16);                                                   // APP0 / JFIF 1.01, no thumbnail, aspect 1:1
    for (uint8_t b : { (uint8_t)'J', (uint8_t)'F', (uint8_t)'I', (uint8_t)'F', (uint8_t)0, (uint8_t)1, (uint8_t)1, (uint8_t)0 }) o.push_back(b);
    be16(1); be16(1); o.push_back(0); o.push_back(0);
    for (int t = 0; t < (channels == 3 ? 2 : 1); ++t) {                       // DQT, zig-zag order
        marker(0xDB); be16(67); o.push_back((uint8_t)t);
        for (int i = 0; i < 64; ++i) o.push_back(q[t][kZigzag[i]]);
    }
    marker(0xC0); be16(8 + 3 * channels); o.push_back(8); be16(h); be16(w); o.push_back((uint8_t)channels);     // SOF0
    for (int c = 0; c < channels; ++c) { o.push_back((uint8_t)(c + 1)); o.push_back(0x11); o.push_back((uint8_t)(c == 0 ? 0 : 1)); }
    auto dht = [&](int cls, int id, const uint8_t* bits, const uint8_t* vals, int nvals) {
        marker(0xC4); be16(19 + nvals); o.push_back((uint8_t)((cls << 4) | id));
        for (int i = 0; i < 16; ++i) o.push_back(bits[i]);
        for (int i = 0; i < nvals; ++i) o.push_back(vals[i]);
    };
    dht(0, 0, kDcLumBits, kDcVals, 12); dht(1, 0, kAcLumBits, kAcLumVals, 162);
    if (channels == 3) { dht(0, 1, kDcChrBits, kDcVals, 12); dht(1, 1, kAcChrBits, kAcChrVals, 162); }
    marker(0xDD); be16(4); be16((w + 7) / 8);                                 // DRI: a restart marker after every row of blocks
    marker(0xDA); be16(6 + 2 * channels); o.push_back((uint8_t)channels);     // SOS
    for (int c = 0; c < channels; ++c) { o.push_back((uint8_t)(c + 1)); o.push_back((uint8_t)(c == 0 ? 0x00 : 0x11)); }
    o.push_back(0); o.push_back(63); o.push_back(0);
}

}  // namespace wassjpeg
