// jpeg.hpp -- a baseline JPEG writer for the debug pictures, so that they carry the reference's file names
// (cv::imwrite("....jpg")).  Sequential DCT, 8-bit, Huffman coding with the typical tables of
// ITU-T T.81 Annex K, quantisation tables of Annex K scaled for quality 95 the way libjpeg scales them (OpenCV's default
// quality), no chroma subsampling (4:4:4), JFIF header.  Grey (1 component) or RGB (3 components, converted to YCbCr with
// the JFIF equations).  The bytes differ from what OpenCV / libjpeg would write for the same picture (their DCT is an
// integer approximation and OpenCV subsamples chroma); a decoder shows the same picture.
#pragma once

#include <cmath>
#include <cstdint>
#include <fstream>
#include <string>
#include <vector>

namespace wasshost {

namespace jpegdetail {

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

struct BitWriter {
    std::vector<uint8_t>& out;
    uint32_t acc = 0;
    int n = 0;
    explicit BitWriter(std::vector<uint8_t>& o) : out(o) {}
    void put(uint32_t bits, int len)
    {
        acc = (acc << len) | (bits & ((1u << len) - 1u));
        n += len;
        while (n >= 8) {
            const uint8_t b = (uint8_t)(acc >> (n - 8));
            out.push_back(b);
            if (b == 0xFF) out.push_back(0);              // byte stuffing
            n -= 8;
        }
    }
    void flush() { if (n > 0) put(0x7F, 8 - n); }         // pad with ones
};

// forward 8x8 DCT-II of a level-shifted block, separable, in double
inline void fdct8x8(const double in[64], double out[64])
{
    struct Table {                                            // built once, thread-safe (function-local static)
        double v[8][8];
        Table() { for (int k = 0; k < 8; ++k) for (int x = 0; x < 8; ++x) v[k][x] = (k == 0 ? std::sqrt(0.125) : 0.5) * std::cos((2 * x + 1) * k * 3.14159265358979323846 / 16.0); }
    };
    static const Table tab;
    const double (&c)[8][8] = tab.v;
    double tmp[64];
    for (int y = 0; y < 8; ++y)
        for (int k = 0; k < 8; ++k) { double s = 0; for (int x = 0; x < 8; ++x) s += c[k][x] * in[y * 8 + x]; tmp[y * 8 + k] = s; }
    for (int k = 0; k < 8; ++k)
        for (int u = 0; u < 8; ++u) { double s = 0; for (int y = 0; y < 8; ++y) s += c[k][y] * tmp[y * 8 + u]; out[k * 8 + u] = s; }
}

inline int bit_size(int v) { int a = v < 0 ? -v : v, n = 0; while (a) { ++n; a >>= 1; } return n; }

}  // namespace jpegdetail

// data: h rows of w pixels, channels = 1 (grey) or 3 (r,g,b interleaved)
inline bool write_jpeg_raw(const std::string& filename, int w, int h, int channels, const uint8_t* data, int quality = 95)
{
    using namespace jpegdetail;
    if (w <= 0 || h <= 0 || w > 65535 || h > 65535 || (channels != 1 && channels != 3)) return false;
    const int scale = quality < 50 ? 5000 / (quality < 1 ? 1 : quality) : 200 - 2 * (quality > 100 ? 100 : quality);
    uint8_t q[2][64];
    for (int t = 0; t < 2; ++t)
        for (int i = 0; i < 64; ++i) {
            int v = ((t == 0 ? kQLum[i] : kQChr[i]) * scale + 50) / 100;
            q[t][i] = (uint8_t)(v < 1 ? 1 : (v > 255 ? 255 : v));
        }
    std::vector<uint8_t> o;
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
    marker(0xDA); be16(6 + 2 * channels); o.push_back((uint8_t)channels);     // SOS
    for (int c = 0; c < channels; ++c) { o.push_back((uint8_t)(c + 1)); o.push_back((uint8_t)(c == 0 ? 0x00 : 0x11)); }
    o.push_back(0); o.push_back(63); o.push_back(0);

    const Huff hdc[2] = { make_huff(kDcLumBits, kDcVals), make_huff(kDcChrBits, kDcVals) };
    const Huff hac[2] = { make_huff(kAcLumBits, kAcLumVals), make_huff(kAcChrBits, kAcChrVals) };
    BitWriter bw(o);
    int pred[3] = { 0, 0, 0 };
    for (int by = 0; by < h; by += 8)
        for (int bx = 0; bx < w; bx += 8)
            for (int c = 0; c < channels; ++c) {
                double blk[64], coef[64];
                for (int y = 0; y < 8; ++y)
                    for (int x = 0; x < 8; ++x) {
                        const int yy = by + y < h ? by + y : h - 1, xx = bx + x < w ? bx + x : w - 1;   // edge blocks: replicate
                        const uint8_t* p = data + ((size_t)yy * w + xx) * channels;
                        double v;
                        if (channels == 1) v = p[0];
                        else if (c == 0) v = 0.299 * p[0] + 0.587 * p[1] + 0.114 * p[2];
                        else if (c == 1) v = -0.168735892 * p[0] - 0.331264108 * p[1] + 0.5 * p[2] + 128.0;
                        else v = 0.5 * p[0] - 0.418687589 * p[1] - 0.081312411 * p[2] + 128.0;
                        blk[y * 8 + x] = v - 128.0;
                    }
                fdct8x8(blk, coef);
                const int t = c == 0 ? 0 : 1;
                int zz[64];
                for (int i = 0; i < 64; ++i) {
                    const long v = std::lrint(coef[kZigzag[i]] / q[t][kZigzag[i]]);
                    zz[i] = (int)(i == 0 ? v : (v < -1023 ? -1023 : (v > 1023 ? 1023 : v)));       // baseline AC amplitudes have at most 10 bits
                }
                const int diff = zz[0] - pred[c];
                pred[c] = zz[0];
                int s = bit_size(diff);
                bw.put(hdc[t].code[s], hdc[t].len[s]);
                if (s) bw.put((uint32_t)(diff < 0 ? diff - 1 : diff), s);
                int run = 0;
                for (int i = 1; i < 64; ++i) {
                    if (zz[i] == 0) { ++run; continue; }
                    while (run > 15) { bw.put(hac[t].code[0xF0], hac[t].len[0xF0]); run -= 16; }
                    s = bit_size(zz[i]);
                    bw.put(hac[t].code[(run << 4) | s], hac[t].len[(run << 4) | s]);
                    bw.put((uint32_t)(zz[i] < 0 ? zz[i] - 1 : zz[i]), s);
                    run = 0;
                }
                if (run) bw.put(hac[t].code[0x00], hac[t].len[0x00]);                               // EOB
            }
    bw.flush();
    marker(0xD9);                                                             // EOI
    std::ofstream ofs(filename.c_str(), std::ios::binary);
    if (ofs.fail()) return false;
    ofs.write((const char*)o.data(), (std::streamsize)o.size());
    return !ofs.fail();
}

}  // namespace wasshost
