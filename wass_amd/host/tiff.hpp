// tiff.hpp -- a reader for the TIFF files cameras and wasscli hand to wass_prepare (cli/wasscli/wasscli.py:47 lists tif / tiff
// among the supported inputs; the reference reads them with cv::imread(IMREAD_GRAYSCALE), wass_prepare.cpp:90).
// Baseline TIFF 6.0, strips (no tiles), chunky planar configuration, 8 or 16 bits per sample, 1 (grey), 3 or 4 samples per
// pixel; compression none (1), LZW (5), Deflate (8 / 32946) or PackBits (32773); horizontal predictor (2); both byte
// orders.  Result: 8-bit grey like cv::imread(IMREAD_GRAYSCALE) -- 16-bit samples keep their high byte, RGB goes through
// the BT.601 fixed-point weights of read_png_gray, WhiteIsZero is inverted.  Anything else is rejected with a message.
#pragma once

#include <zlib.h>

#include "jpeg_read.hpp"

#include <cstdint>
#include <cstring>
#include <fstream>
#include <iterator>
#include <stdexcept>
#include <string>
#include <vector>

namespace wasshost {

namespace tiffdetail {

struct Reader {
    const std::vector<uint8_t>& f;
    bool be;
    uint16_t u16(size_t o) const { if (o + 2 > f.size()) throw std::runtime_error("truncated TIFF"); return be ? (uint16_t)((f[o] << 8) | f[o + 1]) : (uint16_t)(f[o] | (f[o + 1] << 8)); }
    uint32_t u32(size_t o) const
    {
        if (o + 4 > f.size()) throw std::runtime_error("truncated TIFF");
        return be ? ((uint32_t)f[o] << 24) | ((uint32_t)f[o + 1] << 16) | ((uint32_t)f[o + 2] << 8) | f[o + 3]
                  : ((uint32_t)f[o + 3] << 24) | ((uint32_t)f[o + 2] << 16) | ((uint32_t)f[o + 1] << 8) | f[o];
    }
};

// values of an IFD entry of type BYTE / SHORT / LONG
inline std::vector<uint32_t> values(const Reader& r, size_t entry)
{
    const uint16_t type = r.u16(entry + 2);
    const uint32_t count = r.u32(entry + 4);
    const size_t sz = type == 1 ? 1 : (type == 3 ? 2 : (type == 4 ? 4 : 0));
    if (!sz) throw std::runtime_error("unsupported TIFF field type");
    if (count > (1u << 24)) throw std::runtime_error("unreasonable TIFF field count");
    const size_t at = sz * count <= 4 ? entry + 8 : r.u32(entry + 8);
    std::vector<uint32_t> v(count);
    for (uint32_t i = 0; i < count; ++i) v[i] = sz == 1 ? r.f.at(at + i) : (sz == 2 ? r.u16(at + 2 * i) : r.u32(at + 4 * i));
    return v;
}

// TIFF LZW: MSB-first codes of 9..12 bits, ClearCode 256, EndOfInformation 257, the code width grows one code EARLY
inline void lzw_decode(const uint8_t* src, size_t n, std::vector<uint8_t>& out, size_t expect)
{
    struct Entry { int prev; uint8_t ch; uint16_t len; };
    std::vector<Entry> tab(4096);
    for (int i = 0; i < 256; ++i) tab[i] = { -1, (uint8_t)i, 1 };
    int next = 258, width = 9, prev = -1;
    uint32_t acc = 0;
    int bits = 0;
    size_t pos = 0;
    std::vector<uint8_t> tmp;
    auto emit = [&](int code) {
        const size_t len = tab[code].len, base = out.size();
        out.resize(base + len);
        for (int c = code, k = (int)len - 1; c >= 0; c = tab[c].prev, --k) out[base + k] = tab[c].ch;
        return out[base];
    };
    while (out.size() < expect) {
        while (bits < width) { if (pos >= n) return; acc = (acc << 8) | src[pos++]; bits += 8; }
        const int code = (int)((acc >> (bits - width)) & ((1u << width) - 1));
        bits -= width;
        if (code == 257) return;
        if (code == 256) { next = 258; width = 9; prev = -1; continue; }
        if (prev < 0) { if (code >= 256) throw std::runtime_error("corrupt LZW stream"); emit(code); prev = code; continue; }
        uint8_t first;
        if (code < next) first = emit(code);
        else if (code == next) {                          // KwKwK: the string of prev plus its own first byte
            const size_t base = out.size();
            first = emit(prev);
            out.push_back(out[base]);
        } else throw std::runtime_error("corrupt LZW stream");
        if (next < 4096) { tab[next] = { prev, first, (uint16_t)(tab[prev].len + 1) }; ++next; }
        if (next + 1 >= (1 << width) && width < 12) ++width;
        prev = code;
    }
}

inline void packbits_decode(const uint8_t* src, size_t n, std::vector<uint8_t>& out, size_t expect)
{
    size_t p = 0;
    while (p < n && out.size() < expect) {
        const int8_t c = (int8_t)src[p++];
        if (c >= 0) { const size_t k = (size_t)c + 1; if (p + k > n) throw std::runtime_error("corrupt PackBits stream"); out.insert(out.end(), src + p, src + p + k); p += k; }
        else if (c != -128) { if (p >= n) throw std::runtime_error("corrupt PackBits stream"); out.insert(out.end(), (size_t)(1 - c), src[p++]); }
    }
}

}  // namespace tiffdetail

inline bool is_tiff(const std::vector<uint8_t>& f)
{
    return f.size() >= 8 && ((f[0] == 'I' && f[1] == 'I' && f[2] == 42 && f[3] == 0) || (f[0] == 'M' && f[1] == 'M' && f[2] == 0 && f[3] == 42));
}

inline Image decode_tiff_gray(const std::vector<uint8_t>& f, const std::string& name)
{
    using namespace tiffdetail;
    if (!is_tiff(f)) throw std::runtime_error(name + " is not a TIFF file");
    const Reader r{ f, f[0] == 'M' };
    const size_t ifd = r.u32(4);
    const int nent = r.u16(ifd);
    uint32_t w = 0, h = 0, comp = 1, photo = 1, spp = 1, rps = 0xFFFFFFFFu, planar = 1, predictor = 1, bits = 1;
    std::vector<uint32_t> offs, counts, bps;
    bool tiled = false;
    for (int i = 0; i < nent; ++i) {
        const size_t e = ifd + 2 + (size_t)12 * i;
        const uint16_t tag = r.u16(e);
        switch (tag) {
            case 256: w = values(r, e).at(0); break;
            case 257: h = values(r, e).at(0); break;
            case 258: bps = values(r, e); break;
            case 259: comp = values(r, e).at(0); break;
            case 262: photo = values(r, e).at(0); break;
            case 273: offs = values(r, e); break;
            case 277: spp = values(r, e).at(0); break;
            case 278: rps = values(r, e).at(0); break;
            case 279: counts = values(r, e); break;
            case 284: planar = values(r, e).at(0); break;
            case 317: predictor = values(r, e).at(0); break;
            case 322: case 323: case 324: case 325: tiled = true; break;
            default: break;
        }
    }
    if (!bps.empty()) { bits = bps[0]; for (uint32_t b : bps) if (b != bits) throw std::runtime_error(name + ": mixed sample depths are not supported"); }
    if (tiled) throw std::runtime_error(name + ": tiled TIFF files are not supported");
    if (w == 0 || h == 0 || w > 65535 || h > 65535) throw std::runtime_error(name + ": bad TIFF dimensions");
    if ((bits != 8 && bits != 16) || (spp != 1 && spp != 3 && spp != 4) || planar != 1 || photo > 2 || (predictor != 1 && predictor != 2))
        throw std::runtime_error(name + ": only 8/16-bit grey or RGB(A), chunky, strip-organised TIFF files are supported");
    if (comp != 1 && comp != 5 && comp != 8 && comp != 32946 && comp != 32773) throw std::runtime_error(name + ": unsupported TIFF compression " + std::to_string(comp));
    if (offs.empty() || offs.size() != counts.size()) throw std::runtime_error(name + ": missing strip table");
    if (rps == 0xFFFFFFFFu || rps > h) rps = h;
    const size_t bpsamp = bits / 8, rowbytes = (size_t)w * spp * bpsamp;
    Image img((int)w, (int)h);
    std::vector<uint8_t> strip;
    for (size_t s = 0; s < offs.size(); ++s) {
        const size_t y0 = s * rps;
        if (y0 >= h) break;
        const size_t rows = std::min<size_t>(rps, h - y0), expect = rows * rowbytes;
        if ((size_t)offs[s] + counts[s] > f.size()) throw std::runtime_error(name + ": strip outside the file");
        const uint8_t* src = &f[offs[s]];
        strip.clear();
        if (comp == 1) strip.assign(src, src + std::min<size_t>(counts[s], expect));
        else if (comp == 5) { strip.reserve(expect); lzw_decode(src, counts[s], strip, expect); }
        else if (comp == 32773) { strip.reserve(expect); packbits_decode(src, counts[s], strip, expect); }
        else {
            strip.resize(expect);
            uLongf got = (uLongf)expect;
            if (uncompress(strip.data(), &got, src, (uLong)counts[s]) != Z_OK) throw std::runtime_error(name + ": zlib inflate failed");
            strip.resize(got);
        }
        if (strip.size() < expect) throw std::runtime_error(name + ": strip " + std::to_string(s) + " is short");
        for (size_t y = 0; y < rows; ++y) {
            uint8_t* row = &strip[y * rowbytes];
            if (predictor == 2) {                         // horizontal differencing, per sample, in the sample's width
                if (bits == 8) { for (size_t i = spp; i < rowbytes; ++i) row[i] = (uint8_t)(row[i] + row[i - spp]); }
                else for (size_t i = spp; i < (size_t)w * spp; ++i) {
                    const size_t a = 2 * i, b = 2 * (i - spp);
                    const uint16_t cur = r.be ? (uint16_t)((row[a] << 8) | row[a + 1]) : (uint16_t)(row[a] | (row[a + 1] << 8));
                    const uint16_t prv = r.be ? (uint16_t)((row[b] << 8) | row[b + 1]) : (uint16_t)(row[b] | (row[b + 1] << 8));
                    const uint16_t v = (uint16_t)(cur + prv);
                    if (r.be) { row[a] = (uint8_t)(v >> 8); row[a + 1] = (uint8_t)v; } else { row[a] = (uint8_t)v; row[a + 1] = (uint8_t)(v >> 8); }
                }
            }
            for (uint32_t x = 0; x < w; ++x) {
                auto sample = [&](uint32_t c) -> int {
                    const uint8_t* p = row + ((size_t)x * spp + c) * bpsamp;
                    return bits == 8 ? p[0] : (r.be ? p[0] : p[1]);          // 16 bit -> its high byte
                };
                int v = spp == 1 ? sample(0) : ((sample(0) * 4899 + sample(1) * 9617 + sample(2) * 1868 + 8192) >> 14);
                if (photo == 0) v = 255 - v;
                img.at((int)(y0 + y), (int)x) = (uint8_t)v;
            }
        }
    }
    return img;
}

// any supported picture -> 8-bit grey (cv::imread(..., IMREAD_GRAYSCALE)): PNG, TIFF or baseline JPEG, told apart by their magic bytes
inline Image read_image_gray(const std::string& filename)
{
    std::ifstream ifs(filename.c_str(), std::ios::binary);
    if (!ifs.is_open()) throw std::runtime_error("unable to open " + filename);
    std::vector<uint8_t> f((std::istreambuf_iterator<char>(ifs)), std::istreambuf_iterator<char>());
    if (is_tiff(f)) return decode_tiff_gray(f, filename);
    if (is_jpeg(f)) return decode_jpeg_gray(f, filename);
    return read_png_gray(filename);
}

}  // namespace wasshost
