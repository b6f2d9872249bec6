// render.hpp -- the debug pictures wass_stereo leaves in the workdir:
//   stereo.jpg                     wass_stereo.cpp:1910-1925   rectified pair side by side, ROI rectangles, a line every 20 rows
//   stereo_input.jpg               :833                        the two zero-padded SGBM inputs, stacked
//   disparity_stereo_ouput.jpg     :854    (sic)               render_disparity_float of the converted raw disparity
//   disparity_final_scaled.jpg     :1001                       render_disparity_float of the final map
//   disparity_coverage.jpg         :1002-1017                  right image, green = 100 where disparity > 1, ROI, half size
//   graph_components.jpg           PovMesh.cpp:222-250,982-984 biggest component green, the rest in palette colours, half size
//   undistorted/R0.jpg, R1.jpg     :1111-1119,1216-1382        grey where a point was triangulated, else the colour of the rejecting test
//   disparity_large_gradient.jpg, disparity_biggest_component.jpg  :958-960, 981-983   the two masks of the component option
// Same names, same pixel arithmetic (render.hpp:101-136 for the disparity pictures), written by the baseline JPEG encoder of
// jpeg.hpp (quality 95 like cv::imwrite's default; the BYTES differ from libjpeg's, the pictures do not).
// WASS_DEBUG_FORMAT=png writes lossless <stem>.png instead (the tests check the pixel arithmetic on those).
#pragma once

#include <cstdlib>
#include <cstring>

#include "hostio.hpp"
#include "jpeg.hpp"

namespace wasshost {

struct ImageRGB {
    int w = 0, h = 0;
    std::vector<uint8_t> px;     // r,g,b interleaved
    ImageRGB() {}
    ImageRGB(int w_, int h_) : w(w_), h(h_), px((size_t)w_ * h_ * 3, 0) {}
    void set(int y, int x, uint8_t r, uint8_t g, uint8_t b) { if (x >= 0 && y >= 0 && x < w && y < h) { uint8_t* p = &px[((size_t)y * w + x) * 3]; p[0] = r; p[1] = g; p[2] = b; } }
};

inline bool write_png_raw(const std::string& filename, int w, int h, int channels, const uint8_t* data)
{
    const size_t row = (size_t)w * channels;
    std::vector<uint8_t> raw((row + 1) * h);
    for (int y = 0; y < h; ++y) { raw[(row + 1) * y] = 0; memcpy(&raw[(row + 1) * y + 1], data + row * y, row); }
    uLongf clen = compressBound((uLong)raw.size());
    std::vector<uint8_t> comp(clen);
    if (compress2(comp.data(), &clen, raw.data(), (uLong)raw.size(), 1) != Z_OK) return false;
    std::ofstream ofs(filename.c_str(), std::ios::binary);
    if (ofs.fail()) return false;
    auto chunk = [&](const char* type, const uint8_t* d, uint32_t len) {
        uint8_t hdr[8] = { (uint8_t)(len >> 24), (uint8_t)(len >> 16), (uint8_t)(len >> 8), (uint8_t)len, (uint8_t)type[0], (uint8_t)type[1], (uint8_t)type[2], (uint8_t)type[3] };
        ofs.write((const char*)hdr, 8);
        if (len) ofs.write((const char*)d, len);
        uLong crc = crc32(0L, (const Bytef*)type, 4);
        if (len) crc = crc32(crc, d, len);
        const uint8_t c[4] = { (uint8_t)(crc >> 24), (uint8_t)(crc >> 16), (uint8_t)(crc >> 8), (uint8_t)crc };
        ofs.write((const char*)c, 4);
    };
    static const uint8_t sig[8] = { 0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a };
    ofs.write((const char*)sig, 8);
    uint8_t ihdr[13] = { (uint8_t)(w >> 24), (uint8_t)(w >> 16), (uint8_t)(w >> 8), (uint8_t)w, (uint8_t)(h >> 24), (uint8_t)(h >> 16), (uint8_t)(h >> 8), (uint8_t)h,
                         8, (uint8_t)(channels == 3 ? 2 : 0), 0, 0, 0 };
    chunk("IHDR", ihdr, 13);
    chunk("IDAT", comp.data(), (uint32_t)clen);
    chunk("IEND", nullptr, 0);
    return !ofs.fail();
}
inline bool write_png_rgb(const std::string& f, const ImageRGB& im) { return write_png_raw(f, im.w, im.h, 3, im.px.data()); }

// debug pictures: <stem>.jpg like the reference, or <stem>.png with WASS_DEBUG_FORMAT=png
// (per thread: a writer thread of the resident worker finishes frames of callers whose environments differ -- DebugFormatScope)
inline int& debug_png_override() { static thread_local int v = -1; return v; }
inline bool debug_png()
{
    if (debug_png_override() >= 0) return debug_png_override() != 0;
    const char* e = getenv("WASS_DEBUG_FORMAT");
    return e && !strcmp(e, "png");
}
struct DebugFormatScope {
    int saved;
    explicit DebugFormatScope(int v) : saved(debug_png_override()) { if (v >= 0) debug_png_override() = v; }
    ~DebugFormatScope() { debug_png_override() = saved; }
};
inline bool write_debug_gray(const std::string& stem, const Image& im)
{
    return debug_png() ? write_png_gray(stem + ".png", im) : write_jpeg_raw(stem + ".jpg", im.w, im.h, 1, im.px.data());
}
inline bool write_debug_rgb(const std::string& stem, const ImageRGB& im)
{
    return debug_png() ? write_png_rgb(stem + ".png", im) : write_jpeg_raw(stem + ".jpg", im.w, im.h, 3, im.px.data());
}

// render.hpp:101-136: (v - min) / (max - min) * 255 with min starting at cols + 1 and max at 0
inline Image render_disparity_float(const float* disp, int w, int h)
{
    Image out(w, h);
    float mn = (float)(w + 1), mx = 0.0f;
    for (size_t i = 0; i < (size_t)w * h; ++i) { mn = std::min(disp[i], mn); mx = std::max(disp[i], mx); }
    if (!(mx > mn)) return out;                              // an empty map: black (the reference divides by zero here; csrc/jpeg.hip does the same as this)
    for (size_t i = 0; i < (size_t)w * h; ++i) out.px[i] = (unsigned char)((disp[i] - mn) / (mx - mn) * 255.0f);
    return out;
}

inline ImageRGB gray_to_rgb(const Image& g)
{
    ImageRGB o(g.w, g.h);
    for (size_t i = 0; i < g.px.size(); ++i) o.px[3 * i] = o.px[3 * i + 1] = o.px[3 * i + 2] = g.px[i];
    return o;
}
// a crop pasted at (x, y) of a black w x h canvas: the full-size rectified image, of which only the ROI is computed here
inline Image paste(const Image& crop, int x, int y, int w, int h)
{
    Image o(w, h);
    for (int r = 0; r < crop.h; ++r)
        if (y + r >= 0 && y + r < h) memcpy(&o.px[(size_t)(y + r) * w + std::max(x, 0)], &crop.px[(size_t)r * crop.w], (size_t)std::max(0, std::min(crop.w, w - x)));
    return o;
}
// cv::rectangle(img, r, CV_RGB(255,0,0), 3): the outline through the corner pixels, three pixels wide
inline void rectangle_red(ImageRGB& im, int x, int y, int w, int h)
{
    for (int t = -1; t <= 1; ++t) {
        for (int i = x - 1; i <= x + w; ++i) { im.set(y + t, i, 255, 0, 0); im.set(y + h - 1 + t, i, 255, 0, 0); }
        for (int j = y - 1; j <= y + h; ++j) { im.set(j, x + t, 255, 0, 0); im.set(j, x + w - 1 + t, 255, 0, 0); }
    }
}
// cv::resize(..., 0.5, 0.5, INTER_LINEAR) on 8-bit data: every output pixel is the rounded mean of a 2 x 2 block
inline ImageRGB half_size(const ImageRGB& s)
{
    ImageRGB o((s.w + 1) / 2, (s.h + 1) / 2);
    for (int y = 0; y < o.h; ++y)
        for (int x = 0; x < o.w; ++x)
            for (int c = 0; c < 3; ++c) {
                const int x1 = std::min(2 * x + 1, s.w - 1), y1 = std::min(2 * y + 1, s.h - 1);
                const int v = s.px[((size_t)(2 * y) * s.w + 2 * x) * 3 + c] + s.px[((size_t)(2 * y) * s.w + x1) * 3 + c] +
                              s.px[((size_t)y1 * s.w + 2 * x) * 3 + c] + s.px[((size_t)y1 * s.w + x1) * 3 + c];
                o.px[((size_t)y * o.w + x) * 3 + c] = (uint8_t)((v + 2) >> 2);
            }
    return o;
}

}  // namespace wasshost
