// jpeg.hpp -- a baseline JPEG writer for the debug pictures, so that they carry the reference's file names
// (cv::imwrite("....jpg")).  Sequential DCT, 8-bit, Huffman coding with the typical tables of
// ITU-T T.81 Annex K, quantisation tables of Annex K scaled for quality 95 the way libjpeg scales them (OpenCV's default
// quality), no chroma subsampling (4:4:4), JFIF header.  Grey (1 component) or RGB (3 components, converted to YCbCr with
// the JFIF equations in 16-bit fixed point).  The bytes differ from what OpenCV / libjpeg would write for the same picture
// (OpenCV subsamples chroma, libjpeg writes no restart markers); a decoder shows the same picture.
//
// Everything here is INTEGER arithmetic with a defined result -- jpeg_spec.h, shared with the device encoder
// (wass_amd/csrc/jpeg.hip), which writes the same bytes from device-resident pictures: colour conversion, the 8 x 8 forward
// DCT (the 13-bit Loeffler-Ligtenberg-Moshovitz factorisation), quantisation with rounding away from zero at .5, and a
// restart interval of one row of blocks (DRI / RSTn), which is what lets a GPU code the rows side by side.
#pragma once

#include <cstdint>
#include <fstream>
#include <string>
#include <vector>

#include "../csrc/jpeg_spec.h"

namespace wasshost {

// data: h rows of w pixels, channels = 1 (grey) or 3 (r,g,b interleaved).  The file image is appended to `o`.
inline bool encode_jpeg(std::vector<uint8_t>& o, int w, int h, int channels, const uint8_t* data, int quality = 95)
{
    using namespace wassjpeg;
    if (w <= 0 || h <= 0 || w > 65535 || h > 65535 || (channels != 1 && channels != 3)) return false;
    uint8_t q[2][64];
    quant_tables(quality, q);
    file_header(o, w, h, channels, quality);
    static const HuffSet hs = make_huff_set();
    const int bw = (w + 7) / 8, bh = (h + 7) / 8;
    o.reserve(o.size() + (size_t)w * h * channels / 2 + 4096);
    uint64_t acc = 0;                                    // bits not yet written, right-aligned
    int nacc = 0;
    auto put = [&](uint32_t bits, int len) {
        acc = (acc << len) | (bits & ((1u << len) - 1u));
        nacc += len;
        while (nacc >= 8) {
            const uint8_t b = (uint8_t)(acc >> (nacc - 8));
            o.push_back(b);
            if (b == 0xFF) o.push_back(0);               // byte stuffing
            nacc -= 8;
        }
    };
    for (int by = 0; by < bh; ++by) {
        int pred[3] = { 0, 0, 0 };                        // DC prediction starts again in every restart interval
        for (int bx = 0; bx < bw; ++bx)
            for (int c = 0; c < channels; ++c) {
                int blk[64];
                for (int y = 0; y < 8; ++y) {
                    const int yy = by * 8 + y < h ? by * 8 + y : h - 1;                  // edge blocks: replicate
                    for (int x = 0; x < 8; ++x) {
                        const int xx = bx * 8 + x < w ? bx * 8 + x : w - 1;
                        const uint8_t* p = data + ((size_t)yy * w + xx) * channels;
                        blk[y * 8 + x] = (channels == 1 ? p[0] : ycc(p[0], p[1], p[2], c)) - 128;
                    }
                }
                fdct8x8(blk);
                const int t = c == 0 ? 0 : 1;
                int zz[64];
                for (int i = 0; i < 64; ++i) zz[i] = quantise(blk[kZigzag[i]], q[t][kZigzag[i]], i == 0);
                const int diff = zz[0] - pred[c];
                pred[c] = zz[0];
                int s = bit_size(diff);
                put(hs.t[t].code[s], hs.t[t].len[s]);
                if (s) put((uint32_t)(diff < 0 ? diff - 1 : diff), s);
                const Huff& ac = hs.t[2 + t];
                int run = 0;
                for (int i = 1; i < 64; ++i) {
                    if (zz[i] == 0) { ++run; continue; }
                    while (run > 15) { put(ac.code[0xF0], ac.len[0xF0]); run -= 16; }
                    s = bit_size(zz[i]);
                    put(ac.code[(run << 4) | s], ac.len[(run << 4) | s]);
                    put((uint32_t)(zz[i] < 0 ? zz[i] - 1 : zz[i]), s);
                    run = 0;
                }
                if (run) put(ac.code[0x00], ac.len[0x00]);                               // EOB
            }
        if (nacc > 0) put(0x7F, 8 - nacc);                                                // pad the interval with ones
        if (by + 1 < bh) { o.push_back(0xFF); o.push_back((uint8_t)(0xD0 + (by & 7))); }   // RSTm
    }
    o.push_back(0xFF); o.push_back(0xD9);                                                 // EOI
    return true;
}

inline bool write_jpeg_raw(const std::string& filename, int w, int h, int channels, const uint8_t* data, int quality = 95)
{
    std::vector<uint8_t> o;
    if (!encode_jpeg(o, w, h, channels, data, quality)) return false;
    std::ofstream ofs(filename.c_str(), std::ios::binary);
    if (ofs.fail()) return false;
    ofs.write((const char*)o.data(), (std::streamsize)o.size());
    return !ofs.fail();
}

}  // namespace wasshost
