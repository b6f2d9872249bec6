// jpeg_read.hpp -- a reader for baseline JPEG camera frames handed to wass_prepare (wasscli lists jpg / jpeg among its input
// formats, cli/wasscli/wasscli.py:47; the reference reads them with cv::imread(IMREAD_GRAYSCALE), which lets libjpeg decode
// straight to grey, i.e. the luminance component alone).  Sequential DCT (SOF0 / SOF1 with Huffman coding), 8-bit samples,
// 1 or 3 components, any sampling factors for the chroma components (they are parsed and skipped), restart intervals.
// Progressive and arithmetic-coded files are refused with a message.  The inverse DCT is done in double precision; libjpeg's
// default integer IDCT can differ from it by one grey level.
#pragma once

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "hostio.hpp"

namespace wasshost {

namespace jpegrd {

static const uint8_t kZigzag[64] = { 0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
                                     35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63 };

struct HuffTable {
    bool present = false;
    // canonical decoding (ITU-T T.81 F.2.2.3): per length the smallest code, the largest code and the index of its first symbol
    int mincode[17], maxcode[18], valptr[17];
    uint8_t vals[256];
};

struct BitReader {
    const std::vector<uint8_t>& f;
    size_t pos;
    uint32_t acc = 0;
    int n = 0;
    bool hit_marker = false;
    BitReader(const std::vector<uint8_t>& f_, size_t p) : f(f_), pos(p) {}
    void fill()
    {
        while (n <= 24) {
            uint8_t b = 0;
            if (!hit_marker && pos < f.size()) {
                b = f[pos];
                if (b == 0xFF) {
                    const uint8_t nx = pos + 1 < f.size() ? f[pos + 1] : 0xD9;
                    if (nx == 0) pos += 2;                     // stuffed byte
                    else { hit_marker = true; b = 0; }         // a marker: feed zeros until the caller deals with it
                } else ++pos;
            }
            acc |= (uint32_t)b << (24 - n);
            n += 8;
        }
    }
    int bit() { if (n == 0) fill(); const int v = (int)(acc >> 31); acc <<= 1; --n; return v; }
    int bits(int k) { int v = 0; for (int i = 0; i < k; ++i) v = (v << 1) | bit(); return v; }
    void reset_at_marker() { acc = 0; n = 0; hit_marker = false; }
};

inline int decode_symbol(BitReader& br, const HuffTable& h)
{
    int code = 0;
    for (int l = 1; l <= 16; ++l) {
        code = (code << 1) | br.bit();
        if (h.maxcode[l] >= 0 && code <= h.maxcode[l] && code >= h.mincode[l]) return h.vals[h.valptr[l] + code - h.mincode[l]];
    }
    throw std::runtime_error("corrupt JPEG entropy data");
}
inline int extend(int v, int t) { return t == 0 ? 0 : (v < (1 << (t - 1)) ? v - (1 << t) + 1 : v); }

inline void idct8x8(const double in[64], uint8_t* out, size_t stride, int wlim, int hlim)
{
    struct Table {                                            // built once, thread-safe (function-local static)
        double v[8][8];
        Table() { for (int k = 0; k < 8; ++k) for (int x = 0; x < 8; ++x) v[k][x] = (k == 0 ? std::sqrt(0.125) : 0.5) * std::cos((2 * x + 1) * k * 3.14159265358979323846 / 16.0); }
    };
    static const Table tab;
    const double (&c)[8][8] = tab.v;
    double tmp[64];
    for (int v = 0; v < 8; ++v)
        for (int x = 0; x < 8; ++x) { double s = 0; for (int u = 0; u < 8; ++u) s += c[u][x] * in[v * 8 + u]; tmp[v * 8 + x] = s; }
    for (int y = 0; y < hlim; ++y)
        for (int x = 0; x < wlim; ++x) {
            double s = 0;
            for (int v = 0; v < 8; ++v) s += c[v][y] * tmp[v * 8 + x];
            const long r = std::lrint(s + 128.0);
            out[(size_t)y * stride + x] = (uint8_t)(r < 0 ? 0 : (r > 255 ? 255 : r));
        }
}

}  // namespace jpegrd

inline bool is_jpeg(const std::vector<uint8_t>& f) { return f.size() >= 4 && f[0] == 0xFF && f[1] == 0xD8 && f[2] == 0xFF; }

inline Image decode_jpeg_gray(const std::vector<uint8_t>& f, const std::string& name)
{
    using namespace jpegrd;
    if (!is_jpeg(f)) throw std::runtime_error(name + " is not a JPEG file");
    uint16_t qt[4][64] = {};
    HuffTable hdc[4], hac[4];
    struct Comp { int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0, pred = 0; } comp[4];
    int ncomp = 0, W = 0, H = 0, restart = 0;
    size_t p = 2;
    auto u16 = [&](size_t o) { if (o + 2 > f.size()) throw std::runtime_error(name + ": truncated JPEG"); return (int)((f[o] << 8) | f[o + 1]); };
    // every byte of a marker segment is read through these: nothing is taken from beyond the segment's own end
    auto s8 = [&](size_t o, size_t end) { if (o >= end) throw std::runtime_error(name + ": truncated JPEG segment"); return (int)f[o]; };
    auto s16 = [&](size_t o, size_t end) { if (o + 2 > end) throw std::runtime_error(name + ": truncated JPEG segment"); return (int)((f[o] << 8) | f[o + 1]); };
    for (;;) {
        if (p + 4 > f.size()) throw std::runtime_error(name + ": no scan in JPEG file");
        if (f[p] != 0xFF) { ++p; continue; }
        const uint8_t m = f[p + 1];
        if (m == 0xFF) { ++p; continue; }
        p += 2;
        if (m == 0xD8 || (m >= 0xD0 && m <= 0xD7) || m == 0x01) continue;
        const int len = u16(p);
        const size_t seg = p + 2, end = p + len;
        if (len < 2 || end > f.size()) throw std::runtime_error(name + ": truncated JPEG segment");
        if (m == 0xDB) {                                         // DQT
            for (size_t q = seg; q < end;) {
                const int pq = s8(q, end) >> 4, tq = f[q] & 15; ++q;
                if (tq > 3 || pq > 1) throw std::runtime_error(name + ": bad quantisation table");
                for (int i = 0; i < 64; ++i) { qt[tq][kZigzag[i]] = (uint16_t)(pq ? s16(q, end) : s8(q, end)); q += pq ? 2 : 1; }
            }
        } else if (m == 0xC4) {                                  // DHT
            for (size_t q = seg; q < end;) {
                const int tc = s8(q, end) >> 4, th = f[q] & 15; ++q;
                if (tc > 1 || th > 3) throw std::runtime_error(name + ": bad Huffman table");
                HuffTable& h = tc ? hac[th] : hdc[th];
                int counts[17], total = 0;
                for (int l = 1; l <= 16; ++l) { counts[l] = s8(q++, end); total += counts[l]; }
                if (total > 256 || q + total > end) throw std::runtime_error(name + ": bad Huffman table");
                for (int i = 0; i < total; ++i) h.vals[i] = f[q++];
                int code = 0, k = 0;
                for (int l = 1; l <= 16; ++l) {
                    h.valptr[l] = k; h.mincode[l] = code;
                    code += counts[l]; k += counts[l];
                    h.maxcode[l] = counts[l] ? code - 1 : -1;
                    code <<= 1;
                }
                h.present = true;
            }
        } else if (m == 0xC0 || m == 0xC1) {                     // SOF0 / SOF1: sequential, Huffman
            if (len < 8) throw std::runtime_error(name + ": truncated JPEG frame header");
            if (s8(seg, end) != 8) throw std::runtime_error(name + ": only 8-bit JPEG files are supported");
            H = s16(seg + 1, end); W = s16(seg + 3, end); ncomp = s8(seg + 5, end);
            if ((ncomp != 1 && ncomp != 3) || W <= 0 || H <= 0) throw std::runtime_error(name + ": unsupported JPEG layout");
            if (len < 8 + 3 * ncomp) throw std::runtime_error(name + ": truncated JPEG frame header");
            for (int i = 0; i < ncomp; ++i) {
                comp[i].id = s8(seg + 6 + 3 * i, end); comp[i].h = s8(seg + 7 + 3 * i, end) >> 4; comp[i].v = f[seg + 7 + 3 * i] & 15;
                comp[i].tq = s8(seg + 8 + 3 * i, end);
                if (comp[i].tq > 3) throw std::runtime_error(name + ": bad quantisation table");
            }
        } else if (m == 0xC2 || (m >= 0xC5 && m <= 0xCF && m != 0xC8 && m != 0xCC)) {
            throw std::runtime_error(name + ": progressive / lossless / arithmetic-coded JPEG files are not supported (baseline ones are)");
        } else if (m == 0xDD) restart = s16(seg, end);
        else if (m == 0xDA) {                                    // SOS
            if (!W) throw std::runtime_error(name + ": scan before frame header");
            const int ns = s8(seg, end);
            if (ns != ncomp) throw std::runtime_error(name + ": multi-scan JPEG files are not supported");
            for (int i = 0; i < ns; ++i) {
                const int cid = s8(seg + 1 + 2 * i, end), tt = s8(seg + 2 + 2 * i, end);
                if ((tt >> 4) > 3 || (tt & 15) > 3) throw std::runtime_error(name + ": bad Huffman table");       // only four tables of each class exist
                for (int k = 0; k < ncomp; ++k) if (comp[k].id == cid) { comp[k].td = tt >> 4; comp[k].ta = tt & 15; }
            }
            p = end;
            break;
        }
        p = end;
    }
    int hmax = 1, vmax = 1;
    for (int i = 0; i < ncomp; ++i) { if (comp[i].h < 1 || comp[i].v < 1 || comp[i].h > 4 || comp[i].v > 4) throw std::runtime_error(name + ": bad sampling factors"); hmax = std::max(hmax, comp[i].h); vmax = std::max(vmax, comp[i].v); }
    if (ncomp == 1) { comp[0].h = comp[0].v = 1; hmax = vmax = 1; }          // a single-component scan is not interleaved
    if (comp[0].h != hmax || comp[0].v != vmax) throw std::runtime_error(name + ": luminance is subsampled: not supported");
    const int mcuw = 8 * hmax, mcuh = 8 * vmax, mx = (W + mcuw - 1) / mcuw, my = (H + mcuh - 1) / mcuh;
    Image img(W, H);
    BitReader br(f, p);
    int until_restart = restart;
    for (int my_i = 0; my_i < my; ++my_i)
        for (int mx_i = 0; mx_i < mx; ++mx_i) {
            if (restart && until_restart == 0) {                 // RSTn: byte-align, skip the marker, reset the predictors
                br.reset_at_marker();
                while (br.pos + 1 < f.size() && !(f[br.pos] == 0xFF && f[br.pos + 1] >= 0xD0 && f[br.pos + 1] <= 0xD7)) ++br.pos;
                br.pos += 2;
                for (int i = 0; i < ncomp; ++i) comp[i].pred = 0;
                until_restart = restart;
            }
            for (int ci = 0; ci < ncomp; ++ci) {
                Comp& c = comp[ci];
                if (!hdc[c.td].present || !hac[c.ta].present) throw std::runtime_error(name + ": missing Huffman table");
                for (int by = 0; by < c.v; ++by)
                    for (int bx = 0; bx < c.h; ++bx) {
                        double blk[64] = {};
                        const int t = decode_symbol(br, hdc[c.td]);
                        if (t > 16) throw std::runtime_error(name + ": corrupt JPEG entropy data");     // a DC category, not a byte of a damaged table
                        c.pred += extend(br.bits(t), t);
                        blk[0] = (double)c.pred * qt[c.tq][0];
                        for (int k = 1; k < 64;) {
                            const int rs = decode_symbol(br, hac[c.ta]), r = rs >> 4, s = rs & 15;
                            if (s == 0) { if (r == 15) { k += 16; continue; } break; }
                            k += r;
                            if (k > 63) throw std::runtime_error(name + ": corrupt JPEG block");
                            blk[kZigzag[k]] = (double)extend(br.bits(s), s) * qt[c.tq][kZigzag[k]];
                            ++k;
                        }
                        if (ci != 0) continue;                   // chroma: parsed, not reconstructed
                        const int x0 = mx_i * mcuw + bx * 8, y0 = my_i * mcuh + by * 8;
                        if (x0 >= W || y0 >= H) continue;
                        idct8x8(blk, &img.px[(size_t)y0 * W + x0], (size_t)W, std::min(8, W - x0), std::min(8, H - y0));
                    }
            }
            if (restart) --until_restart;
        }
    return img;
}

}  // namespace wasshost
