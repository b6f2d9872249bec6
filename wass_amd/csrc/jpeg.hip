// jpeg.hip -- the reference's debug pictures (wass_stereo.cpp:833, 854, 1001-1017, 1381-1382,
// 1910-1925, PovMesh.cpp:982-984) rendered AND JPEG-coded on the GPU, from the maps the frame chain has in HBM anyway.
//
// Why: the reference writes eight pictures per frame unconditionally (cv::imwrite).  Drawn and coded on the host they cost
// 0.9 s of CPU per 5-megapixel frame -- 4 frames/s through the resident worker where the frame chain itself does 90 -- and
// every intermediate map had to come back to the host first, which took the pipelined chain down to one frame.  Here a
// picture never exists as pixels: the DCT kernel samples a per-picture functor (the arithmetic of host/render.hpp and
// host/wass_frame.hpp, pixel by pixel), and the bytes that cross PCIe are the files'.
//
// The encoder is baseline JPEG with the arithmetic of jpeg_spec.h (shared with host/jpeg.hpp: same bytes, tested) and a
// restart interval of one row of blocks, so that nothing but a prefix sum is sequential:
//   k_jpeg_dct<Src>     one thread per 8x8 block and component: sample, colour-convert, DCT, quantise -> zig-zag int16 coefficients,
//                       the block's AC bit count and its DC value
//   k_jpeg_rows         one workgroup per row of blocks (= restart interval): DC differences, exclusive scan of the blocks' bit
//                       counts -> bit offset of every block, bits of the interval
//   k_jpeg_intervals    one workgroup: byte offsets of the intervals (unstuffed; then, second call, stuffed + restart markers)
//   k_jpeg_emit         one thread per block: Huffman-codes its coefficients and ORs the bits into the unstuffed stream at its offset
//   k_jpeg_count_ff     0xFF bytes per interval
//   k_jpeg_stuff        byte stuffing (FF -> FF 00), RSTm / EOI, straight into the destination (pinned host memory when the caller's
//                       buffer is device-accessible)
#include <algorithm>
#include <cstring>
#include <vector>

#include "common.h"

#define WASS_JPEG_FN __host__ __device__ __forceinline__
#include "jpeg_spec.h"

using namespace wass;

namespace {

struct JQuant { uint8_t q[2][64]; };
struct JHuffDev { uint32_t cl[4][256]; };      // code | length << 16 of [DC luma, DC chroma, AC luma, AC chroma]: one load per symbol

// ------------------------------------------------------------------ pixel sources
// sample(x, y, c): component c of the picture's pixel (x, y); grey pictures have one component, colour ones give Y, Cb, Cr of (r, g, b)
struct SrcPlain {                                   // a picture that exists in memory (wass_jpeg_encode_dev)
    const uint8_t* p; size_t pitch; int ch;
    __device__ int sample(int x, int y, int c) const
    {
        const uint8_t* q = p + (size_t)y * pitch + (size_t)x * ch;
        return ch == 1 ? q[0] : wassjpeg::ycc(q[0], q[1], q[2], c);
    }
};

// A crop pasted at (rx, ry) of a black W0 x H0 canvas (render.hpp paste()): the full-size rectified picture of which only the ROI exists
struct Pasted {
    const uint8_t* crop; int cw, ch, rx, ry, W0, H0;
    __device__ int at(int x, int y) const
    {
        const int r = y - ry, x0 = rx > 0 ? rx : 0, n = cw < W0 - rx ? cw : W0 - rx;
        if (r < 0 || r >= ch || x < x0 || x >= x0 + n) return 0;
        return crop[(size_t)r * cw + (x - x0)];
    }
};
// cv::rectangle(img, roi, red, 3) as render.hpp rectangle_red() draws it: is (x, y) on the outline?
__device__ __forceinline__ bool on_rect(int x, int y, int rx, int ry, int rw, int rh)
{
    const int dy0 = y - ry, dy1 = y - (ry + rh - 1), dx0 = x - rx, dx1 = x - (rx + rw - 1);
    const bool hor = ((dy0 >= -1 && dy0 <= 1) || (dy1 >= -1 && dy1 <= 1)) && x >= rx - 1 && x <= rx + rw;
    const bool ver = ((dx0 >= -1 && dx0 <= 1) || (dx1 >= -1 && dx1 <= 1)) && y >= ry - 1 && y <= ry + rh;
    return hor || ver;
}

struct SrcStereo {                                  // stereo.jpg: left | right with their ROI rectangles, a red line every 20 rows
    Pasted l, r; int rl[4], rr[4];
    __device__ int sample(int x, int y, int c) const
    {
        const int W0 = l.W0, half = x >= W0, xl = half ? x - W0 : x;
        const int* roi = half ? rr : rl;
        if (y % 20 == 0 || on_rect(xl, y, roi[0], roi[1], roi[2], roi[3])) return wassjpeg::ycc(255, 0, 0, c);
        const int g = half ? r.at(xl, y) : l.at(xl, y);
        return wassjpeg::ycc(g, g, g, c);
    }
};
struct SrcInputs {                                  // stereo_input.jpg: the zero-padded SGBM inputs, left above right
    const uint8_t* lc; const uint8_t* rc; int cw, ch, xl0, xr0;
    __device__ int sample(int x, int y, int) const
    {
        const bool top = y < ch;
        const int r = top ? y : y - ch, x0 = top ? xl0 : xr0;
        return (x >= x0 && x < x0 + cw) ? (top ? lc : rc)[(size_t)r * cw + (x - x0)] : 0;
    }
};
// clean_and_convert_disparity (wass_stereo.cpp:714-733) of the raw map, dense_scale = 1
__device__ __forceinline__ float conv_disp(int16_t d16, int min_disp, int num_disp, int disp_offset)
{
    const float dval = ((float)d16) / 16.0f;
    return (dval <= (float)min_disp || dval > (float)num_disp) ? 0.0f : (float)((double)(dval + (float)disp_offset) * 1.0);
}
struct SrcDisp {                                    // render_disparity_float (render.hpp:101-136) of the raw (d16) or the final (f) map
    const int16_t* d16; const float* f; const float* mnmx; int w, min_disp, num_disp, disp_offset;
    __device__ float value(size_t i) const { return d16 ? conv_disp(d16[i], min_disp, num_disp, disp_offset) : f[i]; }
    __device__ int sample(int x, int y, int) const
    {
        const float mn = mnmx[0], mx = mnmx[1];
        if (!(mx > mn)) return 0;
        return (int)(unsigned char)((value((size_t)y * w + x) - mn) / (mx - mn) * 255.0f);
    }
};
struct SrcCoverage {                                // disparity_coverage.jpg: right picture, green = 100 where disparity > 1, ROI, half size
    Pasted r; const float* f; int roi[4], W0, H0;
    __device__ void full(int x, int y, int& R, int& G, int& B) const
    {
        if (on_rect(x, y, roi[0], roi[1], roi[2], roi[3])) { R = 255; G = 0; B = 0; return; }
        R = G = B = r.at(x, y);
        const int u = x - roi[0], v = y - roi[1];
        if (u >= 0 && v >= 0 && u < r.cw && v < r.ch && f[(size_t)v * r.cw + u] > 1.0f) G = 100;
    }
    __device__ int sample(int x, int y, int c) const
    {
        const int x1 = 2 * x + 1 < W0 - 1 ? 2 * x + 1 : W0 - 1, y1 = 2 * y + 1 < H0 - 1 ? 2 * y + 1 : H0 - 1;
        int R[4], G[4], B[4];
        full(2 * x, 2 * y, R[0], G[0], B[0]); full(x1, 2 * y, R[1], G[1], B[1]); full(2 * x, y1, R[2], G[2], B[2]); full(x1, y1, R[3], G[3], B[3]);
        return wassjpeg::ycc((R[0] + R[1] + R[2] + R[3] + 2) >> 2, (G[0] + G[1] + G[2] + G[3] + 2) >> 2, (B[0] + B[1] + B[2] + B[3] + 2) >> 2, c);
    }
};
__constant__ uint8_t kCodeRgb[7][3] = { { 0, 0, 0 }, { 0, 0, 0 }, { 0, 255, 255 }, { 255, 255, 0 }, { 0, 255, 0 }, { 0, 0, 255 }, { 255, 0, 0 } };
struct SrcReject {                                  // undistorted/R0.jpg (which = 0), R1.jpg (1): grey where a point was made, else the rejecting test's colour
    const uint8_t* codes; const uint8_t* rc; const uint8_t* lc; const float* f;
    int rl[4], rr[4], lcw, lch, W0, H0, which; float comp;
    __device__ int sample(int x, int y, int c) const
    {
        const int u = x - rr[0], v = y - rr[1];
        if (u < 0 || v < 0 || u >= rr[2] || v >= rr[3]) return wassjpeg::ycc(0, 0, 0, c);
        const uint8_t cd = codes[(size_t)v * rr[2] + u];
        const int code = which ? cd >> 4 : cd & 15;
        if (code == WASS_CODE_GREY) {
            int g;
            if (!which) g = rc[(size_t)v * rr[2] + u];
            else {
                const float xl = (float)((float)(u + rl[0]) - f[(size_t)v * rr[2] + u] + comp);
                const int lx = (int)floorf(xl + 0.5f) - rl[0], ly = y - rl[1];
                g = (lx >= 0 && lx < lcw && ly >= 0 && ly < lch) ? lc[(size_t)ly * lcw + lx] : 0;
            }
            return wassjpeg::ycc(g, g, g, c);
        }
        if (code == WASS_CODE_NONE || code > 6) return wassjpeg::ycc(0, 0, 0, c);
        return wassjpeg::ycc(kCodeRgb[code][0], kCodeRgb[code][1], kCodeRgb[code][2], c);
    }
};
struct SrcComponents {                              // graph_components.jpg: biggest component green, the rest blue, half size
    const uint8_t* codes; const uint8_t* after; int gw, gh;
    __device__ void full(int x, int y, int& G, int& B) const
    {
        const size_t i = (size_t)y * gw + x;
        G = B = 0;
        if (after[i]) G = 255;
        else if (codes[i] == (WASS_CODE_GREY | (WASS_CODE_GREY << 4))) B = 255;
    }
    __device__ int sample(int x, int y, int c) const
    {
        const int x1 = 2 * x + 1 < gw - 1 ? 2 * x + 1 : gw - 1, y1 = 2 * y + 1 < gh - 1 ? 2 * y + 1 : gh - 1;
        int G[4], B[4];
        full(2 * x, 2 * y, G[0], B[0]); full(x1, 2 * y, G[1], B[1]); full(2 * x, y1, G[2], B[2]); full(x1, y1, G[3], B[3]);
        return wassjpeg::ycc(0, (G[0] + G[1] + G[2] + G[3] + 2) >> 2, (B[0] + B[1] + B[2] + B[3] + 2) >> 2, c);
    }
};

// ------------------------------------------------------------------ encoder kernels
// meta[id] = AC bits << 16 | (DC & 0xffff); blocks in stream order: id = ((by * bw + bx) * C + c)
template <class Src>
__global__ void __launch_bounds__(64) k_jpeg_dct(Src src, int w, int h, int C, int bw, int nblk, JQuant q, const JHuffDev* hf, int16_t* __restrict__ coef,
                                                  uint32_t* __restrict__ meta)
{
    const int id = blockIdx.x * 64 + threadIdx.x;
    if (id >= nblk) return;
    const int c = id % C, m = id / C, bx = m % bw, by = m / bw;
    int blk[64];
#pragma unroll
    for (int y = 0; y < 8; ++y) {
        const int yy = by * 8 + y < h ? by * 8 + y : h - 1;
#pragma unroll
        for (int x = 0; x < 8; ++x) {
            const int xx = bx * 8 + x < w ? bx * 8 + x : w - 1;
            blk[y * 8 + x] = src.sample(xx, yy, c) - 128;
        }
    }
    wassjpeg::fdct8x8(blk);
    const int t = c == 0 ? 0 : 1;
    const uint32_t* accl = hf->cl[2 + t];
    int16_t* out = coef + ((size_t)(id >> 6) << 12) + (id & 63);            // groups of 64 blocks, coefficient-major: out[i * 64] (coalesced)
    int bits = 0, run = 0, dc = 0;
    constexpr uint8_t ZZ[64] = { 0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
                                 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63 };
#pragma unroll
    for (int i = 0; i < 64; ++i) {
        const int v = wassjpeg::quantise(blk[ZZ[i]], q.q[t][ZZ[i]], i == 0);
        out[i * 64] = (int16_t)v;
        if (i == 0) { dc = v; continue; }
        if (v == 0) { ++run; continue; }
        bits += (run >> 4) * (int)(accl[0xF0] >> 16);
        const int s = wassjpeg::bit_size(v);
        bits += (int)(accl[((run & 15) << 4) | s] >> 16) + s;
        run = 0;
    }
    if (run) bits += (int)(accl[0x00] >> 16);
    meta[id] = ((uint32_t)bits << 16) | ((uint32_t)dc & 0xffffu);
}

__device__ __forceinline__ int block_excl_scan(int v, int* total, int* lds)         // 256 threads; returns the exclusive prefix, *total = the sum
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o, 64); if (lane >= o) incl += t; }
    if (lane == 63) lds[wv] = incl;
    __syncthreads();
    int base = 0, tot = 0;
    for (int k = 0; k < 4; ++k) { const int s = lds[k]; if (k < wv) base += s; tot += s; }
    __syncthreads();
    *total = tot;
    return base + incl - v;
}

// one workgroup per restart interval (row of blocks): bit offset of each block inside the interval, bits of the interval
__global__ void __launch_bounds__(256) k_jpeg_rows(const uint32_t* __restrict__ meta, int n_row, int C, const JHuffDev* hf, uint32_t* __restrict__ bitoff,
                                                    uint32_t* __restrict__ ibits)
{
    __shared__ int lds[4];
    const size_t base = (size_t)blockIdx.x * n_row;
    int carry = 0;
    for (int i0 = 0; i0 < n_row; i0 += 256) {
        const int i = i0 + threadIdx.x;
        int len = 0;
        if (i < n_row) {
            const uint32_t m = meta[base + i];
            const int dc = (int)(int16_t)(m & 0xffffu), prev = i >= C ? (int)(int16_t)(meta[base + i - C] & 0xffffu) : 0;
            const int s = wassjpeg::bit_size(dc - prev);
            len = (int)(m >> 16) + (int)(hf->cl[(i % C) == 0 ? 0 : 1][s] >> 16) + s;
        }
        int tot;
        const int ex = block_excl_scan(len, &tot, lds);
        if (i < n_row) bitoff[base + i] = (uint32_t)(carry + ex);
        carry += tot;
    }
    if (threadIdx.x == 0) ibits[blockIdx.x] = (uint32_t)carry;
}

// one workgroup.  pass 0: ibase[i] = sum of ceil(ibits / 8) of the intervals before i (byte offsets in the unstuffed stream), info[0] = their total.
// pass 1: obase[i] = ibase[i] + 0xFF bytes before i + 2 i (a restart marker after every interval), info[1] = the size of the coded data incl. EOI,
// info[2] = 1 if that does not fit into `capacity`
__global__ void __launch_bounds__(256) k_jpeg_intervals(int pass, int nint, const uint32_t* __restrict__ ibits, uint32_t* __restrict__ ibase, const uint32_t* __restrict__ ff,
                                                         uint32_t* __restrict__ obase, uint32_t* __restrict__ info, uint32_t capacity)
{
    __shared__ int lds[4];
    int carry = 0;
    for (int i0 = 0; i0 < nint; i0 += 256) {
        const int i = i0 + threadIdx.x;
        const int v = i < nint ? (pass == 0 ? (int)((ibits[i] + 7) >> 3) : (int)ff[i]) : 0;
        int tot;
        const int ex = block_excl_scan(v, &tot, lds);
        if (i < nint) { if (pass == 0) ibase[i] = (uint32_t)(carry + ex); else obase[i] = ibase[i] + (uint32_t)(carry + ex) + 2u * (uint32_t)i; }
        carry += tot;
    }
    if (threadIdx.x == 0) {
        if (pass == 0) info[0] = (uint32_t)carry;
        else { const uint32_t total = info[0] + (uint32_t)carry + 2u * (uint32_t)nint; info[1] = total; info[2] = total > capacity ? 1u : 0u; }
    }
}

struct BitSink {
    uint32_t* w; unsigned long long acc; int n;
    __device__ void put(uint32_t bits, int len)
    {
        acc = (acc << len) | (unsigned long long)(bits & ((1u << len) - 1u));
        n += len;
        if (n >= 32) {
            atomicOr(w++, __builtin_bswap32((uint32_t)(acc >> (n - 32))));
            n -= 32;
            acc &= (1ull << n) - 1ull;
        }
    }
    __device__ void flush() { if (n > 0) atomicOr(w, __builtin_bswap32((uint32_t)(acc << (32 - n)))); }
};

// one thread per block: its bits into the (zeroed) unstuffed stream U at byte ibase[interval], bit bitoff[block]
__global__ void __launch_bounds__(64) k_jpeg_emit(const int16_t* __restrict__ coef, const uint32_t* __restrict__ meta, const uint32_t* __restrict__ bitoff,
                                                   const uint32_t* __restrict__ ibase, const uint32_t* __restrict__ ibits, int n_row, int C, int nblk, const JHuffDev* hf,
                                                   uint32_t* __restrict__ U, const uint32_t* __restrict__ info, uint32_t u_capacity)
{
    const int id = blockIdx.x * 64 + threadIdx.x;
    if (id >= nblk || info[0] + 4u > u_capacity) return;                      // (the stream would not fit: the picture is reported as too large)
    const int row = id / n_row, i = id - row * n_row, c = i % C, t = c == 0 ? 0 : 1;
    const unsigned long long P = (unsigned long long)ibase[row] * 8ull + bitoff[id];
    BitSink bs{ U + (P >> 5), 0ull, (int)(P & 31) };
    const int16_t* z = coef + ((size_t)(id >> 6) << 12) + (id & 63);
    const int dc = z[0], prev = i >= C ? (int)(int16_t)(meta[id - C] & 0xffffu) : 0, diff = dc - prev;
    int s = wassjpeg::bit_size(diff);
    { const uint32_t e = hf->cl[t][s]; bs.put(e & 0xffffu, (int)(e >> 16)); }
    if (s) bs.put((uint32_t)(diff < 0 ? diff - 1 : diff), s);
    const uint32_t* cl = hf->cl[2 + t];
    int run = 0;
    for (int k = 1; k < 64; ++k) {
        const int v = z[k * 64];
        if (v == 0) { ++run; continue; }
        while (run > 15) { const uint32_t e = cl[0xF0]; bs.put(e & 0xffffu, (int)(e >> 16)); run -= 16; }
        s = wassjpeg::bit_size(v);
        { const uint32_t e = cl[(run << 4) | s]; bs.put(e & 0xffffu, (int)(e >> 16)); }
        bs.put((uint32_t)(v < 0 ? v - 1 : v), s);
        run = 0;
    }
    if (run) { const uint32_t e = cl[0x00]; bs.put(e & 0xffffu, (int)(e >> 16)); }
    if (i == n_row - 1) { const int pad = (int)((8u - (ibits[row] & 7u)) & 7u); if (pad) bs.put((1u << pad) - 1u, pad); }    // the interval ends on a byte: ones
    bs.flush();
}

__global__ void __launch_bounds__(256) k_jpeg_count_ff(const uint8_t* __restrict__ U, const uint32_t* __restrict__ ibase, const uint32_t* __restrict__ ibits,
                                                        uint32_t* __restrict__ ff, const uint32_t* __restrict__ info, uint32_t u_capacity)
{
    __shared__ int lds[4];
    const uint32_t n = (ibits[blockIdx.x] + 7) >> 3;
    const uint8_t* p = U + ibase[blockIdx.x];
    int cnt = 0;
    if (info[0] + 4u <= u_capacity) for (uint32_t j = threadIdx.x; j < n; j += 256) cnt += p[j] == 0xFF;
    int tot;
    (void)block_excl_scan(cnt, &tot, lds);
    if (threadIdx.x == 0) ff[blockIdx.x] = (uint32_t)tot;
}

// byte stuffing and markers, one workgroup per interval, into dst (device memory or device-visible pinned host memory)
__global__ void __launch_bounds__(256) k_jpeg_stuff(const uint8_t* __restrict__ U, const uint32_t* __restrict__ ibase, const uint32_t* __restrict__ ibits,
                                                     const uint32_t* __restrict__ obase, const uint32_t* __restrict__ info, int nint, uint8_t* __restrict__ dst)
{
    __shared__ int lds[4];
    if (info[2]) return;
    const int it = blockIdx.x;
    const uint32_t n = (ibits[it] + 7) >> 3;
    const uint8_t* p = U + ibase[it];
    uint8_t* o = dst + obase[it];
    uint32_t carry = 0;
    for (uint32_t j0 = 0; j0 < n; j0 += 256) {
        const uint32_t j = j0 + threadIdx.x;
        const int b = j < n ? p[j] : 0, f = b == 0xFF;
        int tot;
        const int ex = block_excl_scan(f, &tot, lds);
        if (j < n) { o[j + carry + ex] = (uint8_t)b; if (f) o[j + carry + ex + 1] = 0; }
        carry += (uint32_t)tot;
    }
    if (threadIdx.x == 0) { o[n + carry] = 0xFF; o[n + carry + 1] = it + 1 < nint ? (uint8_t)(0xD0 + (it & 7)) : (uint8_t)0xD9; }
}

// min / max of a float map as render_disparity_float starts them (min = w + 1, max = 0): two stages, exact
template <class V>
__global__ void __launch_bounds__(256) k_minmax_part(V v, size_t n, float init_min, float* __restrict__ part)
{
    __shared__ float smn[256], smx[256];
    float mn = init_min, mx = 0.0f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) { const float x = v.value(i); mn = fminf(x, mn); mx = fmaxf(x, mx); }
    smn[threadIdx.x] = mn; smx[threadIdx.x] = mx;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) { smn[threadIdx.x] = fminf(smn[threadIdx.x], smn[threadIdx.x + s]); smx[threadIdx.x] = fmaxf(smx[threadIdx.x], smx[threadIdx.x + s]); }
        __syncthreads();
    }
    if (threadIdx.x == 0) { part[2 * blockIdx.x] = smn[0]; part[2 * blockIdx.x + 1] = smx[0]; }
}
__global__ void k_minmax_final(const float* __restrict__ part, int nb, float* __restrict__ out)
{
    if (threadIdx.x || blockIdx.x) return;
    float mn = part[0], mx = part[1];
    for (int i = 1; i < nb; ++i) { mn = fminf(mn, part[2 * i]); mx = fmaxf(mx, part[2 * i + 1]); }
    out[0] = mn; out[1] = mx;
}

struct JScratch {
    int16_t* coef; uint32_t *meta, *bitoff, *ibits, *ibase, *ff, *obase, *U; size_t u_cap;
};

// scratch for a picture of nblk blocks in nint intervals whose coded data may take u_cap bytes
int jpeg_scratch(wass_ctx* c, size_t nblk, size_t nint, size_t u_cap, JScratch& s)
{
    int rc;
    const size_t ucap4 = ((u_cap + 15) & ~(size_t)15) + 16;       // (+16: a stream that does not fit here does not fit the destination either)
    const size_t ngrp = (nblk + 63) / 64;
    const size_t need = ngrp * 8192 + nblk * 8 + nint * 20 + ucap4 + 256;
    if ((rc = ensure(c, c->jpeg_scratch, need))) return rc;
    char* p = (char*)c->jpeg_scratch.p;
    s.coef = (int16_t*)p; p += ngrp * 8192;
    s.meta = (uint32_t*)p; p += nblk * 4;
    s.bitoff = (uint32_t*)p; p += nblk * 4;
    s.ibits = (uint32_t*)p; p += nint * 4;
    s.ibase = (uint32_t*)p; p += nint * 4;
    s.ff = (uint32_t*)p; p += nint * 4;
    s.obase = (uint32_t*)p; p += nint * 4;
    p += nint * 4;
    p = (char*)(((uintptr_t)p + 15) & ~(uintptr_t)15);
    s.U = (uint32_t*)p;
    s.u_cap = ucap4;
    return WASS_OK;
}

int jpeg_tables(wass_ctx* c, hipStream_t st)
{
    if (c->jpeg_tables_ready) return WASS_OK;
    int rc;
    if ((rc = ensure(c, c->jpeg_huff, sizeof(JHuffDev)))) return rc;
    static JHuffDev host;                                                   // (static: the copy below is asynchronous)
    const wassjpeg::HuffSet hs = wassjpeg::make_huff_set();
    for (int t = 0; t < 4; ++t) for (int i = 0; i < 256; ++i) host.cl[t][i] = (uint32_t)hs.t[t].code[i] | ((uint32_t)hs.t[t].len[i] << 16);
    WASS_HIP(c, hipMemcpyAsync(c->jpeg_huff.p, &host, sizeof host, hipMemcpyHostToDevice, st));
    c->jpeg_tables_ready = true;
    return WASS_OK;
}

// The coded data of one picture (everything after the SOS header, EOI included) into dst[0 .. capacity); info[1] = its size, info[2] = too large.
// All on stream st; info is three device-visible words.
template <class Src>
int encode_picture(wass_ctx* c, hipStream_t st, const Src& src, int w, int h, int C, int quality, uint8_t* dst, uint32_t capacity, uint32_t* info)
{
    int rc;
    if ((rc = jpeg_tables(c, st))) return rc;
    const int bw = (w + 7) / 8, bh = (h + 7) / 8, n_row = bw * C, nblk = n_row * bh;
    JScratch s;
    if ((rc = jpeg_scratch(c, (size_t)nblk, (size_t)bh, capacity, s))) return rc;
    JQuant q;
    wassjpeg::quant_tables(quality, q.q);
    const JHuffDev* hf = (const JHuffDev*)c->jpeg_huff.p;
    WASS_HIP(c, hipMemsetAsync(s.U, 0, s.u_cap, st));
    hipLaunchKernelGGL(k_jpeg_dct<Src>, dim3((nblk + 63) / 64), dim3(64), 0, st, src, w, h, C, bw, nblk, q, hf, s.coef, s.meta);
    hipLaunchKernelGGL(k_jpeg_rows, dim3(bh), dim3(256), 0, st, (const uint32_t*)s.meta, n_row, C, hf, s.bitoff, s.ibits);
    hipLaunchKernelGGL(k_jpeg_intervals, dim3(1), dim3(256), 0, st, 0, bh, (const uint32_t*)s.ibits, s.ibase, (const uint32_t*)s.ff, s.obase, info, capacity);
    hipLaunchKernelGGL(k_jpeg_emit, dim3((nblk + 63) / 64), dim3(64), 0, st, (const int16_t*)s.coef, (const uint32_t*)s.meta, (const uint32_t*)s.bitoff,
                       (const uint32_t*)s.ibase, (const uint32_t*)s.ibits, n_row, C, nblk, hf, s.U, (const uint32_t*)info, (uint32_t)s.u_cap);
    hipLaunchKernelGGL(k_jpeg_count_ff, dim3(bh), dim3(256), 0, st, (const uint8_t*)s.U, (const uint32_t*)s.ibase, (const uint32_t*)s.ibits, s.ff, (const uint32_t*)info,
                       (uint32_t)s.u_cap);
    hipLaunchKernelGGL(k_jpeg_intervals, dim3(1), dim3(256), 0, st, 1, bh, (const uint32_t*)s.ibits, s.ibase, (const uint32_t*)s.ff, s.obase, info, capacity);
    hipLaunchKernelGGL(k_jpeg_stuff, dim3(bh), dim3(256), 0, st, (const uint8_t*)s.U, (const uint32_t*)s.ibase, (const uint32_t*)s.ibits, (const uint32_t*)s.obase,
                       (const uint32_t*)info, bh, dst);
    WASS_HIP(c, hipGetLastError());
    return WASS_OK;
}

template <class V>
int minmax(wass_ctx* c, hipStream_t st, const V& v, size_t n, float init_min, float* out)
{
    int rc;
    const int nb = 512;
    if ((rc = ensure(c, c->jpeg_part, nb * 8))) return rc;
    hipLaunchKernelGGL(k_minmax_part<V>, dim3(nb), dim3(256), 0, st, v, n, init_min, (float*)c->jpeg_part.p);
    hipLaunchKernelGGL(k_minmax_final, dim3(1), dim3(64), 0, st, (const float*)c->jpeg_part.p, nb, out);
    return WASS_OK;
}

uint8_t* device_view(void* host_ptr)
{
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, host_ptr) == hipSuccess && at.type == hipMemoryTypeHost && at.devicePointer) return (uint8_t*)at.devicePointer;
    (void)hipGetLastError();
    return nullptr;
}

}  // namespace

// ------------------------------------------------------------------ C ABI
extern "C" {

int wass_jpeg_encode_dev(wass_ctx* c, const uint8_t* d_pixels, int w, int h, int channels, size_t pitch_bytes, int quality, uint8_t* h_dst, size_t capacity,
                         size_t* nbytes)
{
    if (!c || !d_pixels || !h_dst || !nbytes) return set_err(c, WASS_ERR_INVALID_ARG, "null argument");
    if (w <= 0 || h <= 0 || w > 65535 || h > 65535 || (channels != 1 && channels != 3) || pitch_bytes < (size_t)w * channels)
        return set_err(c, WASS_ERR_INVALID_ARG, "bad picture geometry %d x %d x %d, pitch %zu", w, h, channels, pitch_bytes);
    WASS_HIP(c, hipSetDevice(c->device));
    std::vector<uint8_t> hdr;
    wassjpeg::file_header(hdr, w, h, channels, quality);
    if (capacity < hdr.size() + 4) return set_err(c, WASS_ERR_INVALID_ARG, "capacity %zu too small", capacity);
    const size_t room = std::min<size_t>(capacity - hdr.size(), 0xfffffff0u);
    int rc;
    if ((rc = ensure(c, c->jpeg_out, room + 16))) return rc;
    if ((rc = ensure(c, c->jpeg_info, 64))) return rc;
    hipStream_t st = c->ts();
    uint32_t* info = (uint32_t*)c->jpeg_info.p;
    WASS_HIP(c, hipMemsetAsync(info, 0, 64, st));
    SrcPlain src{ d_pixels, pitch_bytes, channels };
    if ((rc = encode_picture(c, st, src, w, h, channels, quality, (uint8_t*)c->jpeg_out.p, (uint32_t)room, info))) return rc;
    uint32_t hinfo[4] = {};
    WASS_HIP(c, hipMemcpyAsync(hinfo, info, 16, hipMemcpyDeviceToHost, st));
    WASS_HIP(c, hipStreamSynchronize(st));
    if (hinfo[2]) return set_err(c, WASS_ERR_INVALID_ARG, "the coded picture (%u bytes) does not fit into %zu", hinfo[1], room);
    memcpy(h_dst, hdr.data(), hdr.size());
    WASS_HIP(c, hipMemcpy(h_dst + hdr.size(), c->jpeg_out.p, hinfo[1], hipMemcpyDeviceToHost));
    *nbytes = hdr.size() + hinfo[1];
    return WASS_OK;
}

static void picture_geometry(const wass_debug_desc* d, int k, int* w, int* h, int* C)
{
    const int W0 = d->W0, H0 = d->H0, cw = d->roi_r[2], ch = d->roi_r[3], offp = d->disp_offset > 0 ? d->disp_offset : 0;
    switch (k) {
        case WASS_PIC_STEREO: *w = 2 * W0; *h = H0; *C = 3; break;
        case WASS_PIC_STEREO_INPUT: *w = cw + d->num_disp + offp; *h = 2 * ch; *C = 1; break;
        case WASS_PIC_DISPARITY_RAW: case WASS_PIC_DISPARITY_FINAL: *w = cw; *h = ch; *C = 1; break;
        case WASS_PIC_COVERAGE: *w = (W0 + 1) / 2; *h = (H0 + 1) / 2; *C = 3; break;
        case WASS_PIC_R0: case WASS_PIC_R1: *w = W0; *h = H0; *C = 3; break;
        default: *w = (cw + 1) / 2; *h = (ch + 1) / 2; *C = 3; break;
    }
}

int wass_debug_picture_size(const wass_debug_desc* d, int k, int* width, int* height, int* channels)
{
    if (!d || k < 0 || k >= WASS_DEBUG_PICTURES || !width || !height || !channels) return WASS_ERR_INVALID_ARG;
    picture_geometry(d, k, width, height, channels);
    return WASS_OK;
}

// The eight debug pictures of a frame, enqueued behind its tail.  h_dst must be pinned host memory (wass_pinned_alloc): the coded bytes are
// written into it by the kernels.  Layout: picture k (wass_gpu.h) as a complete file at h_dst + offset[k], offset[0] = 0,
// offset[k + 1] = offset[k] + capacity[k] rounded up to 64; sizes (0 = not written: too large for its slot) through
// wass_debug_pictures_result once the ticket's work is done.
int wass_debug_pictures_async(wass_ctx* c, const wass_mesh* m, const wass_debug_desc* d, uint8_t* h_dst, const size_t capacity[WASS_DEBUG_PICTURES], uint64_t* ticket)
{
    if (!c || !m || !d || !h_dst || !capacity || !ticket) return set_err(c, WASS_ERR_INVALID_ARG, "null argument");
    WASS_HIP(c, hipSetDevice(c->device));
    const int W0 = d->W0, H0 = d->H0, cw = d->roi_r[2], ch = d->roi_r[3];
    if (W0 <= 0 || H0 <= 0 || cw <= 0 || ch <= 0 || d->roi_l[2] != cw || d->roi_l[3] != ch || m->w != cw || m->h != ch || !m->codes)
        return set_err(c, WASS_ERR_INVALID_ARG, "debug pictures: the two ROIs and the mesh must have one size");
    if (!d->d_left_crop || !d->d_right_crop || !d->d_disp16 || !d->d_dispf || !c->ccmask.p || c->ccmask.cap < (size_t)cw * ch)
        return set_err(c, WASS_ERR_INVALID_ARG, "debug pictures: a map is missing (the frame tail must have been asked for the component mask)");
    uint8_t* dv = device_view(h_dst);
    if (!dv) return set_err(c, WASS_ERR_INVALID_ARG, "debug pictures: the destination must be pinned host memory");
    const int slot = (int)(c->dbg_tickets % 4);
    if (!c->ev_dbg[slot]) WASS_HIP(c, hipEventCreateWithFlags(&c->ev_dbg[slot], hipEventDisableTiming | hipEventBlockingSync));
    int rc;
    if ((rc = ensure(c, c->jpeg_info, 4 * WASS_DEBUG_PICTURES * 16 + 64))) return rc;
    hipStream_t st = c->ts();
    uint32_t* info = (uint32_t*)c->jpeg_info.p + slot * WASS_DEBUG_PICTURES * 4;       // [picture][4]
    float* mnmx = (float*)((uint32_t*)c->jpeg_info.p + 4 * WASS_DEBUG_PICTURES * 4);     // two pairs
    WASS_HIP(c, hipMemsetAsync(info, 0, WASS_DEBUG_PICTURES * 16, st));
    const int q = d->quality > 0 ? d->quality : 95;
    const int D = d->num_disp, offp = d->disp_offset > 0 ? d->disp_offset : 0, comp = d->disp_offset > 0 ? 0 : -d->disp_offset;
    const Pasted pl{ d->d_left_crop, cw, ch, d->roi_l[0], d->roi_l[1], W0, H0 }, pr{ d->d_right_crop, cw, ch, d->roi_r[0], d->roi_r[1], W0, H0 };
    struct Pic { int w, h, C; } pic[WASS_DEBUG_PICTURES];
    size_t offset = 0;
    for (int k = 0; k < WASS_DEBUG_PICTURES; ++k) {
        picture_geometry(d, k, &pic[k].w, &pic[k].h, &pic[k].C);
        std::vector<uint8_t> hdr;
        wassjpeg::file_header(hdr, pic[k].w, pic[k].h, pic[k].C, q);
        if (capacity[k] < hdr.size() + 64 || pic[k].w > 65535 || pic[k].h > 65535)
            return set_err(c, WASS_ERR_INVALID_ARG, "debug pictures: slot of %zu bytes / picture %d x %d", capacity[k], pic[k].w, pic[k].h);
        memcpy(h_dst + offset, hdr.data(), hdr.size());                               // (the caller owns the buffer until it has read the result)
        uint8_t* dst = dv + offset + hdr.size();
        const uint32_t cap = (uint32_t)std::min<size_t>(capacity[k] - hdr.size(), 0xfffffff0u);
        offset += (capacity[k] + 63) & ~(size_t)63;
        uint32_t* inf = info + 4 * k;
        switch (k) {
            case WASS_PIC_STEREO: {
                SrcStereo s{ pl, pr, { d->roi_l[0], d->roi_l[1], d->roi_l[2], d->roi_l[3] }, { d->roi_r[0], d->roi_r[1], d->roi_r[2], d->roi_r[3] } };
                rc = encode_picture(c, st, s, pic[k].w, pic[k].h, 3, q, dst, cap, inf);
                break;
            }
            case WASS_PIC_STEREO_INPUT: {
                SrcInputs s{ d->d_left_crop, d->d_right_crop, cw, ch, D + offp - comp, D };
                rc = encode_picture(c, st, s, pic[k].w, pic[k].h, 1, q, dst, cap, inf);
                break;
            }
            case WASS_PIC_DISPARITY_RAW: case WASS_PIC_DISPARITY_FINAL: {
                const bool raw = k == WASS_PIC_DISPARITY_RAW;
                SrcDisp s{ raw ? d->d_disp16 : nullptr, raw ? nullptr : d->d_dispf, mnmx + (raw ? 0 : 2), cw, d->min_disp, d->num_disp, d->disp_offset };
                if ((rc = minmax(c, st, s, (size_t)cw * ch, (float)(cw + 1), mnmx + (raw ? 0 : 2)))) return rc;
                rc = encode_picture(c, st, s, cw, ch, 1, q, dst, cap, inf);
                break;
            }
            case WASS_PIC_COVERAGE: {
                SrcCoverage s{ pr, d->d_dispf, { d->roi_r[0], d->roi_r[1], d->roi_r[2], d->roi_r[3] }, W0, H0 };
                rc = encode_picture(c, st, s, pic[k].w, pic[k].h, 3, q, dst, cap, inf);
                break;
            }
            case WASS_PIC_R0: case WASS_PIC_R1: {
                SrcReject s{ m->codes, d->d_right_crop, d->d_left_crop, d->d_dispf, { d->roi_l[0], d->roi_l[1], d->roi_l[2], d->roi_l[3] },
                             { d->roi_r[0], d->roi_r[1], d->roi_r[2], d->roi_r[3] }, cw, ch, W0, H0, k == WASS_PIC_R1, (float)(d->disparity_compensation / 1.0) };
                rc = encode_picture(c, st, s, W0, H0, 3, q, dst, cap, inf);
                break;
            }
            default: {
                SrcComponents s{ m->codes, (const uint8_t*)c->ccmask.p, cw, ch };
                rc = encode_picture(c, st, s, pic[k].w, pic[k].h, 3, q, dst, cap, inf);
                break;
            }
        }
        if (rc) return rc;
        c->dbg_hdr[slot][k] = (uint32_t)hdr.size();
    }
    if (!c->h_dbg_info) WASS_HIP(c, hipHostMalloc((void**)&c->h_dbg_info, 4 * WASS_DEBUG_PICTURES * 16, hipHostMallocDefault));
    WASS_HIP(c, hipMemcpyAsync(c->h_dbg_info + slot * WASS_DEBUG_PICTURES * 4, info, WASS_DEBUG_PICTURES * 16, hipMemcpyDeviceToHost, st));
    WASS_HIP(c, hipEventRecord(c->ev_dbg[slot], st));
    *ticket = ++c->dbg_tickets;
    return WASS_OK;
}

int wass_debug_pictures_result(wass_ctx* c, uint64_t ticket, size_t nbytes[WASS_DEBUG_PICTURES])
{
    if (!c || !nbytes) return set_err(c, WASS_ERR_INVALID_ARG, "null argument");
    if (ticket == 0 || ticket > c->dbg_tickets || c->dbg_tickets - ticket >= 4) return set_err(c, WASS_ERR_INVALID_ARG, "debug pictures: ticket %llu is gone", (unsigned long long)ticket);
    WASS_HIP(c, hipSetDevice(c->device));
    const int slot = (int)((ticket - 1) % 4);
    WASS_HIP(c, hipEventSynchronize(c->ev_dbg[slot]));
    const uint32_t* inf = c->h_dbg_info + slot * WASS_DEBUG_PICTURES * 4;
    for (int k = 0; k < WASS_DEBUG_PICTURES; ++k) nbytes[k] = inf[4 * k + 2] ? 0 : (size_t)c->dbg_hdr[slot][k] + inf[4 * k + 1];
    return WASS_OK;
}

}  // extern "C"
