// post.hip -- disparity clean-up after SGBM.
//
// float32 throughout, one thread per pixel, same operation order as the
// reference so that results are bit-identical:
//   clean_and_convert_disparity  wass_stereo/wass_stereo.cpp:714-733
//   matrix_dilate_zero<float>    :617-662  (output column k is filled from the
//                                 8-neighbourhood of column k+1 -- the quirk)
//   matrix_erode_zero<float>     :665-711
//   resize/mask step             :903-928  (same-size copies at DENSE_SCALE 1)
//   cv::medianBlur (3 or 5)      :941-945
#include "common.h"

namespace wass {

__global__ void __launch_bounds__(256) k_convert(const int16_t* __restrict__ d16, size_t n, int mindisp, int numdisp,
                                                 int disp_offset, double scale, float* __restrict__ out)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float dval = ((float)d16[i]) / 16.0f;
    float r = 0.0f;
    if (!(dval <= (float)mindisp || dval > (float)numdisp)) {
        dval += (float)disp_offset;
        r = (float)((double)dval * scale);
    }
    out[i] = r;
}

// DISCARD_BURNED_AREAS (wass_stereo.cpp:1072, 1086): mask = 0 where the undistorted image is saturated (value > 254)
__global__ void __launch_bounds__(256) k_burned_mask(const uint8_t* __restrict__ img, size_t n, uint8_t* __restrict__ mask)
{
    const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i + 4 <= n) {
        const uint32_t v = *(const uint32_t*)(img + i);
        uint32_t m = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) m |= (uint32_t)(((v >> (8 * k)) & 0xFF) <= 254) << (8 * k);
        *(uint32_t*)(mask + i) = m;
    } else {
        for (size_t k = i; k < n; ++k) mask[k] = img[k] <= 254;
    }
}

// the camera masks of triangulate() (wass_stereo.cpp:1057-1093): the thresholded mask picture of the configuration (file:
// 0/1 bytes, may be null) AND, with DISCARD_BURNED_AREAS, "the undistorted picture is not saturated" (img, may be null)
__global__ void __launch_bounds__(256) k_camera_mask(const uint8_t* __restrict__ img, const uint8_t* __restrict__ file, size_t n,
                                                     uint8_t* __restrict__ mask)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    uint8_t m = file ? (file[i] ? 1 : 0) : 1;
    if (img && img[i] > 254) m = 0;
    mask[i] = m;
}

__global__ void __launch_bounds__(256) k_dilate_zero(const float* __restrict__ src, float* __restrict__ out, int w, int h)
{
    const int k = blockIdx.x * 256 + threadIdx.x;
    const int i = blockIdx.y;
    if (k >= w) return;
    const size_t idx = (size_t)i * w + k;
    float v = src[idx];
    // rows 1..h-2, output columns 0..w-3, stencil centred on column k+1
    if (i >= 1 && i < h - 1 && k <= w - 3 && v == 0.0f) {
        const float* t = src + (size_t)(i - 1) * w + (k + 1);
        const float* b = src + (size_t)(i + 1) * w + (k + 1);
        const float* c = src + (size_t)i * w + (k + 1);
        float avg = 0.0f; int n = 0;
        if (t[-1] > 0) { avg += t[-1]; ++n; }
        if (t[1] > 0) { avg += t[1]; ++n; }
        if (t[0] > 0) { avg += t[0]; ++n; }
        if (b[-1] > 0) { avg += b[-1]; ++n; }
        if (b[1] > 0) { avg += b[1]; ++n; }
        if (b[0] > 0) { avg += b[0]; ++n; }
        if (c[-1] > 0) { avg += c[-1]; ++n; }
        if (c[1] > 0) { avg += c[1]; ++n; }
        if (n > 1) v = avg / (float)n;
    }
    out[idx] = v;
}

// MASK: out = (eroded value == 0) ? 0 : keep[idx]   (the NN/cubic step, :908-928)
template <bool MASK>
__global__ void __launch_bounds__(256) k_erode_zero(const float* __restrict__ src, const float* __restrict__ keep,
                                                    float* __restrict__ out, int w, int h)
{
    const int j = blockIdx.x * 256 + threadIdx.x;
    const int i = blockIdx.y;
    if (j >= w) return;
    const size_t idx = (size_t)i * w + j;
    float v = src[idx];
    if (i == 0 || i == h - 1) v = 0.0f;                         // first and last row
    else if (j == 0 || (j == w - 1 && w >= 2)) v = 0.0f;        // first and last column
    else if (j < w - 1) {
        const float* t = src + idx - w;
        const float* b = src + idx + w;
        const float* c = src + idx;
        if (t[0] == 0 || t[-1] == 0 || t[1] == 0 || b[0] == 0 || b[-1] == 0 || b[1] == 0 || c[-1] == 0 || c[1] == 0)
            v = 0.0f;
    }
    out[idx] = MASK ? (v == 0.0f ? 0.0f : keep[idx]) : v;
}

// cv::medianBlur on CV_32F, ksize 3 or 5, replicate border.
template <int KS>
__global__ void __launch_bounds__(256) k_median_f32(const float* __restrict__ src, float* __restrict__ out, int w, int h)
{
    const int x = blockIdx.x * 256 + threadIdx.x;
    const int y = blockIdx.y;
    if (x >= w) return;
    constexpr int R = KS / 2, N = KS * KS;
    float v[N];
#pragma unroll
    for (int a = -R; a <= R; ++a)
#pragma unroll
        for (int b = -R; b <= R; ++b) {
            const int yy = min(max(y + a, 0), h - 1), xx = min(max(x + b, 0), w - 1);
            v[(a + R) * KS + (b + R)] = src[(size_t)yy * w + xx];
        }
    // partial selection: after N/2+1 passes v[N/2] is the median
#pragma unroll
    for (int i = 0; i <= N / 2; ++i)
#pragma unroll
        for (int j = i + 1; j < N; ++j) {
            const float lo = fminf(v[i], v[j]), hi = fmaxf(v[i], v[j]);
            v[i] = lo; v[j] = hi;
        }
    out[(size_t)y * w + x] = v[N / 2];
}

// ---------------------------------------------------------------------------------------------------------------------
// The whole same-size clean-up chain in one launch: clean_and_convert -> dilate x nd -> erode x ne -> mask by one more
// erosion -> optional median, tile by tile in LDS (the separate kernels above move the 20 MB float map through HBM once
// per step).  Every step evaluates the SAME per-pixel expressions as the one-step kernels at global coordinates, so the
// results are bit-identical; a tile carries a halo wide enough that the cells an output pixel depends on are all computed
// inside the tile: one row / column per erosion and per median ring, one row and two columns to the right per dilation
// (its stencil is centred on column k+1).  Cells whose stencil would leave the tile keep their value -- that garbage
// moves inwards by exactly the halo consumed per step and never reaches the output area.
// ---------------------------------------------------------------------------------------------------------------------
struct ChainDims { int w, h, nd, ne, med, hl, hr, hv, tw, th; };

constexpr int CHAIN_TX = 64, CHAIN_TY = 32;

template <int KS>
__device__ __forceinline__ float median_at(const float* __restrict__ buf, int tw, int lx0, int ly0, int x, int y, int w, int h)
{
    constexpr int R = KS / 2, N = KS * KS;
    float v[N];
#pragma unroll
    for (int a = -R; a <= R; ++a)
#pragma unroll
        for (int b = -R; b <= R; ++b) {
            const int yy = min(max(y + a, 0), h - 1), xx = min(max(x + b, 0), w - 1);
            v[(a + R) * KS + (b + R)] = buf[(yy - ly0) * tw + (xx - lx0)];
        }
#pragma unroll
    for (int i = 0; i <= N / 2; ++i)
#pragma unroll
        for (int j = i + 1; j < N; ++j) {
            const float lo = fminf(v[i], v[j]), hi = fmaxf(v[i], v[j]);
            v[i] = lo; v[j] = hi;
        }
    return v[N / 2];
}

__global__ void __launch_bounds__(256) k_clean_chain(const int16_t* __restrict__ d16, ChainDims cd, int mindisp, int numdisp, int disp_offset,
                                                     double scale, float* __restrict__ out)
{
    extern __shared__ float chain_lds[];
    const int tw = cd.tw, th = cd.th, cells = tw * th, w = cd.w, h = cd.h;
    float* A = chain_lds;
    float* B = chain_lds + cells;
    const int gx0 = blockIdx.x * CHAIN_TX - cd.hl, gy0 = blockIdx.y * CHAIN_TY - cd.hv;      // global coordinates of tile cell (0, 0)
    // ---- clean_and_convert (k_convert)
    for (int t = threadIdx.x; t < cells; t += 256) {
        const int ly = t / tw, lx = t - ly * tw, y = gy0 + ly, x = gx0 + lx;
        float r = 0.0f;
        if (x >= 0 && x < w && y >= 0 && y < h) {
            float dval = ((float)d16[(size_t)y * w + x]) / 16.0f;
            if (!(dval <= (float)mindisp || dval > (float)numdisp)) {
                dval += (float)disp_offset;
                r = (float)((double)dval * scale);
            }
        }
        A[t] = r;
    }
    __syncthreads();
    // ---- dilations (k_dilate_zero)
    for (int s = 0; s < cd.nd; ++s) {
        for (int t = threadIdx.x; t < cells; t += 256) {
            const int ly = t / tw, lx = t - ly * tw, i = gy0 + ly, k = gx0 + lx;
            float v = A[t];
            if (i >= 1 && i < h - 1 && k >= 0 && k <= w - 3 && v == 0.0f && ly >= 1 && ly < th - 1 && lx + 2 < tw) {
                const float* tp = A + t - tw + 1;
                const float* bp = A + t + tw + 1;
                const float* cp = A + t + 1;
                float avg = 0.0f; int n = 0;
                if (tp[-1] > 0) { avg += tp[-1]; ++n; }
                if (tp[1] > 0) { avg += tp[1]; ++n; }
                if (tp[0] > 0) { avg += tp[0]; ++n; }
                if (bp[-1] > 0) { avg += bp[-1]; ++n; }
                if (bp[1] > 0) { avg += bp[1]; ++n; }
                if (bp[0] > 0) { avg += bp[0]; ++n; }
                if (cp[-1] > 0) { avg += cp[-1]; ++n; }
                if (cp[1] > 0) { avg += cp[1]; ++n; }
                if (n > 1) v = avg / (float)n;
            }
            B[t] = v;
        }
        __syncthreads();
        float* tmp = A; A = B; B = tmp;
    }
    // ---- erosions (k_erode_zero<false>), the last one as the mask of the map before it (k_erode_zero<true> with nn == cub)
    for (int s = 0; s <= cd.ne; ++s) {
        const bool mask = s == cd.ne;
        for (int t = threadIdx.x; t < cells; t += 256) {
            const int ly = t / tw, lx = t - ly * tw, i = gy0 + ly, j = gx0 + lx;
            const float c0 = A[t];
            float v = c0;
            if (i >= 0 && i < h && j >= 0 && j < w) {
                if (i == 0 || i == h - 1) v = 0.0f;
                else if (j == 0 || (j == w - 1 && w >= 2)) v = 0.0f;
                else if (j < w - 1 && ly >= 1 && ly < th - 1 && lx >= 1 && lx < tw - 1) {
                    const float* tp = A + t - tw;
                    const float* bp = A + t + tw;
                    const bool z = (tp[0] == 0) | (tp[-1] == 0) | (tp[1] == 0) | (bp[0] == 0) | (bp[-1] == 0) | (bp[1] == 0) | (A[t - 1] == 0) |
                                   (A[t + 1] == 0);
                    if (z) v = 0.0f;
                }
            }
            B[t] = mask ? (v == 0.0f ? 0.0f : c0) : v;
        }
        __syncthreads();
        float* tmp = A; A = B; B = tmp;
    }
    // ---- output area (+ median)
    for (int t = threadIdx.x; t < CHAIN_TX * CHAIN_TY; t += 256) {
        const int oy = t / CHAIN_TX, ox = t - oy * CHAIN_TX, y = blockIdx.y * CHAIN_TY + oy, x = blockIdx.x * CHAIN_TX + ox;
        if (x >= w || y >= h) continue;
        float v;
        if (cd.med == 3) v = median_at<3>(A, tw, gx0, gy0, x, y, w, h);
        else if (cd.med == 5) v = median_at<5>(A, tw, gx0, gy0, x, y, w, h);
        else v = A[(oy + cd.hv) * tw + ox + cd.hl];
        out[(size_t)y * w + x] = v;
    }
}

}  // namespace wass

using namespace wass;

extern "C" int wass_disparity_postprocess_ex_dev(wass_ctx* c, const int16_t* d_disp16, int ws, int hs, const wass_sgm_params* p,
                                                 int dilate_steps, int erode_steps, int median_wsize, int cc_threshold, int ow,
                                                 int oh, float* d_out)
{
    if (!c || !d_disp16 || !p || !d_out || ws <= 0 || hs <= 0 || ow <= 0 || oh <= 0) return set_err(c, WASS_ERR_INVALID_ARG, "bad argument");
    if (c->tail_overlap && c->have_last) WASS_HIP(c, hipStreamWaitEvent(c->tail, c->ev[6], 0));   // the SGM call that produced d_disp16
    if (!(p->dense_scale > 0)) return set_err(c, WASS_ERR_INVALID_ARG, "DENSE_SCALE must be positive");
    if (median_wsize >= 3 && median_wsize != 3 && median_wsize != 5)
        return set_err(c, WASS_ERR_UNSUPPORTED, "MEDIAN_FILTER_WSIZE must be 0, 3 or 5 for float maps (cv::medianBlur)");
    WASS_HIP(c, hipSetDevice(c->device));
    const size_t n = (size_t)ws * hs, no = (size_t)ow * oh;
    const bool same = ow == ws && oh == hs;                      // cv::resize to the same size is a copy
    int rc;
    if ((rc = ensure(c, c->fA, (n > no ? n : no) * 4)) || (rc = ensure(c, c->fB, (n > no ? n : no) * 4))) return rc;
    if (!same && ((rc = ensure(c, c->fD, no * 4)) || (rc = ensure(c, c->fE, no * 4)))) return rc;
    float* a = (float*)c->fA.p;
    float* b = (float*)c->fB.p;
    hipStream_t s = c->ts();
    const dim3 grid2((ws + 255) / 256, hs), gout((ow + 255) / 256, oh), blk(256);
    const int off = p->disp_offset > 0 ? p->disp_offset : 0;    // :803-808
    {
        const int rm = median_wsize >= 3 ? median_wsize / 2 : 0;
        ChainDims cd;
        cd.w = ws; cd.h = hs; cd.nd = dilate_steps > 0 ? dilate_steps : 0; cd.ne = erode_steps > 0 ? erode_steps : 0; cd.med = median_wsize >= 3 ? median_wsize : 0;
        cd.hl = cd.ne + 1 + rm; cd.hr = 2 * cd.nd + cd.ne + 1 + rm; cd.hv = cd.nd + cd.ne + 1 + rm;
        cd.tw = CHAIN_TX + cd.hl + cd.hr; cd.th = CHAIN_TY + 2 * cd.hv;
        const size_t lds = (size_t)cd.tw * cd.th * 2 * sizeof(float);
        const char* env = getenv("WASS_CLEAN_CHAIN");
        if (same && lds <= 64 * 1024 && !(env && atoi(env) == 0)) {
            hipLaunchKernelGGL(k_clean_chain, dim3((ws + CHAIN_TX - 1) / CHAIN_TX, (hs + CHAIN_TY - 1) / CHAIN_TY), blk, lds, s, d_disp16, cd,
                               p->min_disp, p->num_disp, off, 1.0 / p->dense_scale, d_out);
            if (c->tail_overlap) WASS_HIP(c, hipEventRecord(c->ev_post, s));
            WASS_HIP(c, hipGetLastError());
            if (cc_threshold > 0) {                              // :947-986
                if ((rc = ensure(c, c->tmp_mask, no))) return rc;
                if ((rc = biggest_component_dev(c, d_out, ow, oh, cc_threshold, (uint8_t*)c->tmp_mask.p, s))) return rc;
            }
            return WASS_OK;
        }
    }
    hipLaunchKernelGGL(k_convert, dim3((unsigned)((n + 255) / 256)), blk, 0, s, d_disp16, n, p->min_disp, p->num_disp,
                       off, 1.0 / p->dense_scale, a);
    // d_disp16 has been consumed: the next SGM call may overwrite it (it waits for this before its last kernel)
    if (c->tail_overlap) WASS_HIP(c, hipEventRecord(c->ev_post, s));
    for (int k = 1; k <= dilate_steps; ++k) {
        hipLaunchKernelGGL(k_dilate_zero, grid2, blk, 0, s, (const float*)a, b, ws, hs);
        float* t = a; a = b; b = t;
    }
    for (int k = 1; k <= erode_steps; ++k) {
        hipLaunchKernelGGL(k_erode_zero<false>, grid2, blk, 0, s, (const float*)a, (const float*)nullptr, b, ws, hs);
        float* t = a; a = b; b = t;
    }
    // :903-928  nearest and bicubic copies at roi_comb_right.size(); out = cubic copy where erode(nearest copy) != 0
    const float* nn = a;
    const float* cub = a;
    if (!same) {
        if ((rc = resize_f32_dev(c, a, ws, hs, (float*)c->fD.p, ow, oh, false, s)) || (rc = resize_f32_dev(c, a, ws, hs, (float*)c->fE.p, ow, oh, true, s)))
            return rc;
        nn = (const float*)c->fD.p; cub = (const float*)c->fE.p;
    }
    float* masked = median_wsize >= 3 ? b : d_out;
    hipLaunchKernelGGL(k_erode_zero<true>, gout, blk, 0, s, nn, cub, masked, ow, oh);
    if (median_wsize == 3) hipLaunchKernelGGL(k_median_f32<3>, gout, blk, 0, s, (const float*)masked, d_out, ow, oh);
    else if (median_wsize == 5) hipLaunchKernelGGL(k_median_f32<5>, gout, blk, 0, s, (const float*)masked, d_out, ow, oh);
    WASS_HIP(c, hipGetLastError());
    if (cc_threshold > 0) {                                      // :947-986
        if ((rc = ensure(c, c->tmp_mask, no))) return rc;
        if ((rc = biggest_component_dev(c, d_out, ow, oh, cc_threshold, (uint8_t*)c->tmp_mask.p, s))) return rc;
    }
    return WASS_OK;
}

extern "C" int wass_disparity_postprocess_dev(wass_ctx* c, const int16_t* d_disp16, int w, int h,
                                              const wass_sgm_params* p, int dilate_steps, int erode_steps,
                                              int median_wsize, float* d_out)
{
    if (p && p->dense_scale != 1.0)
        return set_err(c, WASS_ERR_INVALID_ARG, "DENSE_SCALE != 1: the map and the output differ in size, use wass_disparity_postprocess_ex");
    return wass_disparity_postprocess_ex_dev(c, d_disp16, w, h, p, dilate_steps, erode_steps, median_wsize, 0, w, h, d_out);
}

extern "C" int wass_disparity_postprocess_ex(wass_ctx* c, const int16_t* disp16, int ws, int hs, const wass_sgm_params* p,
                                             int dilate_steps, int erode_steps, int median_wsize, int cc_threshold, int ow, int oh,
                                             float* out)
{
    if (!c || !disp16 || !out || ws <= 0 || hs <= 0 || ow <= 0 || oh <= 0) return set_err(c, WASS_ERR_INVALID_ARG, "bad argument");
    WASS_HIP(c, hipSetDevice(c->device));
    const size_t n = (size_t)ws * hs, no = (size_t)ow * oh;
    int rc;
    if ((rc = ensure(c, c->tmp_out, n * 2)) || (rc = ensure(c, c->fC, no * 4))) return rc;
    WASS_HIP(c, hipMemcpyAsync(c->tmp_out.p, disp16, n * 2, hipMemcpyHostToDevice, c->ts()));
    rc = wass_disparity_postprocess_ex_dev(c, (const int16_t*)c->tmp_out.p, ws, hs, p, dilate_steps, erode_steps, median_wsize,
                                           cc_threshold, ow, oh, (float*)c->fC.p);
    if (rc) return rc;
    WASS_HIP(c, hipMemcpyAsync(out, c->fC.p, no * 4, hipMemcpyDeviceToHost, c->ts()));
    WASS_HIP(c, hipStreamSynchronize(c->ts()));
    return WASS_OK;
}

extern "C" int wass_disparity_postprocess(wass_ctx* c, const int16_t* disp16, int w, int h, const wass_sgm_params* p,
                                          int dilate_steps, int erode_steps, int median_wsize, float* out)
{
    if (p && p->dense_scale != 1.0)
        return set_err(c, WASS_ERR_INVALID_ARG, "DENSE_SCALE != 1: the map and the output differ in size, use wass_disparity_postprocess_ex");
    return wass_disparity_postprocess_ex(c, disp16, w, h, p, dilate_steps, erode_steps, median_wsize, 0, w, h, out);
}

extern "C" int wass_burned_area_mask_dev(wass_ctx* c, const uint8_t* d_img, size_t n, uint8_t* d_mask)
{
    if (!c || !d_img || !d_mask) return wass::set_err(c, WASS_ERR_INVALID_ARG, "null argument");
    if (((uintptr_t)d_img | (uintptr_t)d_mask) & 3) return wass::set_err(c, WASS_ERR_INVALID_ARG, "image and mask must be 4-byte aligned");
    WASS_HIP(c, hipSetDevice(c->device));
    if (int rc = wass::wait_uploads(c, d_img, c->stream)) return rc;
    hipLaunchKernelGGL(wass::k_burned_mask, dim3((unsigned)((n / 4 + 256) / 256)), dim3(256), 0, c->stream, d_img, n, d_mask);
    WASS_HIP(c, hipGetLastError());
    return WASS_OK;
}

extern "C" int wass_camera_mask_dev(wass_ctx* c, const uint8_t* d_img, const uint8_t* d_file_mask, size_t n, uint8_t* d_mask)
{
    if (!c || !d_mask || n == 0) return wass::set_err(c, WASS_ERR_INVALID_ARG, "bad argument");
    WASS_HIP(c, hipSetDevice(c->device));
    int rc;
    if (d_img && (rc = wass::wait_uploads(c, d_img, c->stream))) return rc;
    if (d_file_mask && (rc = wass::wait_uploads(c, d_file_mask, c->stream))) return rc;
    hipLaunchKernelGGL(wass::k_camera_mask, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, d_img, d_file_mask, n, d_mask);
    WASS_HIP(c, hipGetLastError());
    return WASS_OK;
}

// Uploads run on the context's copy stream (a DMA engine, no compute units).  Nothing waits for them until a call that
// READS the destination is enqueued (wass_sgm_disparity_dev, wass_burned_area_mask_dev look their inputs up here), so the
// next frame's images, uploaded while a frame is being processed, land during that frame's aggregation instead of
// between two frames (measured: the SGM stream used to idle ~0.2 ms per frame behind the copy).
namespace wass {
int wait_uploads(wass_ctx* c, const void* p, hipStream_t s)
{
    for (auto& u : c->uploads)
        if (u.pending && (const char*)p >= u.dst && (const char*)p < u.dst + u.n) {
            WASS_HIP(c, hipStreamWaitEvent(s, u.ev, 0));       // stays pending: another stream may read the same buffer
            u.consumed = true;
        }
    return WASS_OK;
}
}  // namespace wass

extern "C" int wass_upload_async(wass_ctx* c, void* d_dst, const void* h_src, size_t nbytes)
{
    if (!c || !d_dst || !h_src) return wass::set_err(c, WASS_ERR_INVALID_ARG, "null argument");
    WASS_HIP(c, hipSetDevice(c->device));
    auto& u = c->uploads[c->upload_next];
    c->upload_next = (c->upload_next + 1) % 8;
    if (!u.ev) WASS_HIP(c, hipEventCreateWithFlags(&u.ev, hipEventDisableTiming));
    if (u.pending && !u.consumed) WASS_HIP(c, hipStreamWaitEvent(c->stream, u.ev, 0));     // never consumed: order it conservatively
    WASS_HIP(c, hipMemcpyAsync(d_dst, h_src, nbytes, hipMemcpyHostToDevice, c->copy));
    WASS_HIP(c, hipEventRecord(u.ev, c->copy));
    u.dst = (const char*)d_dst; u.n = nbytes; u.pending = true; u.consumed = false;
    return WASS_OK;
}
