// post_opt.hip -- the optional parts of sgbm_dense_stereo; off in the WASS defaults, so
// these kernels are written for correctness and clarity, not for the last microsecond:
//   DENSE_SCALE != 1     wass_stereo/wass_stereo.cpp:788-796 cv::resize of both crops (INTER_CUBIC), :903-904 cv::resize of the
//                        float disparity to roi_comb_right.size() (INTER_NEAREST + INTER_CUBIC)
//   DENSE_DISPARITY_BIGGEST_COMPONENT_THRESHOLD > 0   :947-986: zero where the squared Sobel gradient exceeds the threshold,
//                        keep the largest 8-connected component of the rest
//   DENSE_SPECKLE_WINDOW_SIZE > 0   cv::filterSpeckles inside cv::StereoSGBM::compute (:758-759,781-782)
// OpenCV is restated as in oracle/a9_oracle.c (scalar forms; parity unpinned): same arithmetic, operation by operation.
#include "common.h"

namespace wass {

__device__ __forceinline__ int iclip(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// Keys cubic kernel, A = -0.75, evaluated in float exactly like cv::interpolateCubic
__device__ __forceinline__ void cubic_axis(int d, double scale, int& s0, float (&c)[4])
{
    float f = (float)(((double)d + 0.5) * scale - 0.5);
    const int s = (int)floorf(f);
    f -= (float)s;
    s0 = s;
    const float A = -0.75f;
    c[0] = ((A * (f + 1) - 5 * A) * (f + 1) + 8 * A) * (f + 1) - 4 * A;
    c[1] = ((A + 2) * f - (A + 3)) * f * f + 1;
    c[2] = ((A + 2) * (1 - f) - (A + 3)) * (1 - f) * (1 - f) + 1;
    c[3] = 1.f - c[0] - c[1] - c[2];
}
__device__ __forceinline__ int coef_q11(float c) { return iclip(__float2int_rn(c * 2048.0f), -32768, 32767); }

// cv::resize INTER_CUBIC, CV_8UC1: 11-bit fixed-point weights, horizontal pass in int, vertical (sum + 2^21) >> 22
__global__ void __launch_bounds__(256) k_resize_cubic_u8(const uint8_t* __restrict__ src, int sw, int sh, size_t pitch,
                                                         uint8_t* __restrict__ dst, int dw, int dh, double scale_x, double scale_y)
{
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= dw) return;
    int sx, sy;
    float cx[4], cy[4];
    cubic_axis(x, scale_x, sx, cx);
    cubic_axis(y, scale_y, sy, cy);
    int acc = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint8_t* row = src + (size_t)iclip(sy - 1 + k, 0, sh - 1) * pitch;
        int hs = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) hs += (int)row[iclip(sx - 1 + j, 0, sw - 1)] * coef_q11(cx[j]);
        acc += hs * coef_q11(cy[k]);
    }
    dst[(size_t)y * dw + x] = (uint8_t)iclip((acc + (1 << 21)) >> 22, 0, 255);
}

// cv::resize of CV_32FC1: MODE 0 INTER_NEAREST, 1 INTER_CUBIC (float, taps accumulated left to right)
template <int MODE>
__global__ void __launch_bounds__(256) k_resize_f32(const float* __restrict__ src, int sw, int sh, float* __restrict__ dst, int dw,
                                                    int dh, double scale_x, double scale_y)
{
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= dw) return;
    if (MODE == 0) {
        const int sy = iclip((int)floor(y * scale_y), 0, sh - 1), sx = iclip((int)floor(x * scale_x), 0, sw - 1);
        dst[(size_t)y * dw + x] = src[(size_t)sy * sw + sx];
    } else {
        int sx, sy;
        float cx[4], cy[4], rows[4];
        cubic_axis(x, scale_x, sx, cx);
        cubic_axis(y, scale_y, sy, cy);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float* row = src + (size_t)iclip(sy - 1 + k, 0, sh - 1) * sw;
            rows[k] = row[iclip(sx - 1, 0, sw - 1)] * cx[0] + row[iclip(sx, 0, sw - 1)] * cx[1] + row[iclip(sx + 1, 0, sw - 1)] * cx[2] +
                      row[iclip(sx + 2, 0, sw - 1)] * cx[3];
        }
        dst[(size_t)y * dw + x] = rows[0] * cy[0] + rows[1] * cy[1] + rows[2] * cy[2] + rows[3] * cy[3];
    }
}

__device__ __forceinline__ int refl101(int i, int n) { if (n == 1) return 0; if (i < 0) i = -i; if (i >= n) i = 2 * n - 2 - i; return iclip(i, 0, n - 1); }

// cv::Sobel x / y (3x3, BORDER_REFLECT_101), squared magnitude > threshold -> flag
__global__ void __launch_bounds__(256) k_large_gradient(const float* __restrict__ disp, int w, int h, float thr, uint8_t* __restrict__ large)
{
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= w) return;
    float t[3], u[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float* r = disp + (size_t)refl101(y - 1 + k, h) * w;
        const float a = r[refl101(x - 1, w)], b = r[x], c = r[refl101(x + 1, w)];
        t[k] = c - a;
        u[k] = a + b * 2 + c;
    }
    const float gx = t[0] + t[1] * 2 + t[2], gy = u[2] - u[0];
    large[(size_t)y * w + x] = (gx * gx + gy * gy) > thr;
}
__global__ void __launch_bounds__(256) k_zero_where(float* __restrict__ disp, size_t n, const uint8_t* __restrict__ flag)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n && flag[i]) disp[i] = 0.0f;
}

// ---- union-find over a w x h grid.  Roots are the smallest raster index of their set, so "the component whose first pixel
// comes first" (the label order of cv::connectedComponents, the seed order of cv::filterSpeckles) is "the smallest root".
__device__ __forceinline__ int uf_find(int* parent, int i)
{
    int r = i;
    while (true) { const int p = parent[r]; if (p == r) break; r = p; }
    // Path compression against concurrent unions: the chain may have changed since the walk above (another thread may have
    // compressed it onto a newer, smaller root), so never touch an index <= r and only ever LOWER a parent -- every parent
    // stays smaller than its child, which is what keeps the forest acyclic.
    while (i > r) { const int p = parent[i]; if (p > r) parent[i] = r; i = p; }
    return r;
}
__device__ __forceinline__ void uf_union(int* parent, int a, int b)
{
    while (true) {
        a = uf_find(parent, a); b = uf_find(parent, b);
        if (a == b) return;
        if (a < b) { const int t = a; a = b; b = t; }                      // a > b: hang a under b
        const int old = atomicMin(&parent[a], b);
        if (old == a) return;
        a = old;                                                            // somebody re-parented a meanwhile: retry from there
    }
}
// KIND 0: float map, active = value != 0, 8-connected.  KIND 1: int16 map, active = value != newVal, 4-connected and
// |difference| <= maxDiff between the two neighbours.
template <int KIND>
__global__ void __launch_bounds__(256) k_uf_init(const void* __restrict__ img, size_t n, int newVal, int* __restrict__ parent,
                                                 unsigned int* __restrict__ size)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const bool act = KIND == 0 ? ((const float*)img)[i] != 0.0f : ((const int16_t*)img)[i] != newVal;
    parent[i] = act ? (int)i : -1;
    size[i] = 0;
}
template <int KIND>
__global__ void __launch_bounds__(256) k_uf_merge(const void* __restrict__ img, int w, int h, int maxDiff, int* __restrict__ parent)
{
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= w) return;
    const int i = y * w + x;
    if (parent[i] < 0) return;
    auto link = [&](int qx, int qy) {
        if (qx < 0 || qy < 0 || qx >= w) return;
        const int q = qy * w + qx;
        if (parent[q] < 0) return;
        if (KIND == 1) {
            const int a = ((const int16_t*)img)[i], b = ((const int16_t*)img)[q];
            if ((a > b ? a - b : b - a) > maxDiff) return;
        }
        uf_union(parent, i, q);
    };
    link(x - 1, y); link(x, y - 1);
    if (KIND == 0) { link(x - 1, y - 1); link(x + 1, y - 1); }
}
__global__ void __launch_bounds__(256) k_uf_count(size_t n, int* __restrict__ parent, unsigned int* __restrict__ size)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n || parent[i] < 0) return;
    const int r = uf_find(parent, (int)i);
    parent[i] = r;
    atomicAdd(&size[r], 1u);
}
// largest area, first label on ties: max of (area << 32) | ~root over the roots
__global__ void __launch_bounds__(256) k_uf_best(size_t n, const int* __restrict__ parent, const unsigned int* __restrict__ size,
                                                 unsigned long long* __restrict__ best)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n || parent[i] != (int)i) return;
    atomicMax(best, ((unsigned long long)size[i] << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned int)i));
}
__global__ void __launch_bounds__(256) k_keep_best(float* __restrict__ disp, size_t n, const int* __restrict__ parent,
                                                   const unsigned long long* __restrict__ best)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const unsigned long long b = *best;
    const int root = b ? (int)(0xFFFFFFFFu - (unsigned int)(b & 0xFFFFFFFFu)) : -2;
    if (parent[i] != root) disp[i] = 0.0f;
}
__global__ void __launch_bounds__(256) k_speckle_apply(int16_t* __restrict__ img, size_t n, const int* __restrict__ parent,
                                                       const unsigned int* __restrict__ size, int maxSize, int newVal)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n || parent[i] < 0) return;
    if (size[parent[i]] <= (unsigned int)maxSize) img[i] = (int16_t)newVal;
}

static int uf_buffers(wass_ctx* c, size_t n, int** parent, unsigned int** size, unsigned long long** best)
{
    int rc = ensure(c, c->uf, n * 8 + 64);
    if (rc) return rc;
    *parent = (int*)c->uf.p;
    *size = (unsigned int*)(*parent + n);
    *best = (unsigned long long*)((char*)c->uf.p + n * 8);
    return WASS_OK;
}

// cv::filterSpeckles(img, newVal, maxSpeckleSize, maxDiff) in place on a device int16 image
int speckle_filter_dev(wass_ctx* c, int16_t* img, int w, int h, int newVal, int maxSize, int maxDiff, hipStream_t s)
{
    const size_t n = (size_t)w * h;
    int* parent; unsigned int* size; unsigned long long* best;
    int rc = uf_buffers(c, n, &parent, &size, &best);
    if (rc) return rc;
    const dim3 g1((unsigned)((n + 255) / 256)), g2((w + 255) / 256, h), blk(256);
    hipLaunchKernelGGL(k_uf_init<1>, g1, blk, 0, s, (const void*)img, n, newVal, parent, size);
    hipLaunchKernelGGL(k_uf_merge<1>, g2, blk, 0, s, (const void*)img, w, h, maxDiff, parent);
    hipLaunchKernelGGL(k_uf_count, g1, blk, 0, s, n, parent, size);
    hipLaunchKernelGGL(k_speckle_apply, g1, blk, 0, s, img, n, (const int*)parent, (const unsigned int*)size, maxSize, newVal);
    WASS_HIP(c, hipGetLastError());
    return WASS_OK;
}

// wass_stereo.cpp:947-986 in place on a device float map; flag = scratch of n bytes
int biggest_component_dev(wass_ctx* c, float* disp, int w, int h, int threshold, uint8_t* flag, hipStream_t s)
{
    const size_t n = (size_t)w * h;
    int* parent; unsigned int* size; unsigned long long* best;
    int rc = uf_buffers(c, n, &parent, &size, &best);
    if (rc) return rc;
    const dim3 g1((unsigned)((n + 255) / 256)), g2((w + 255) / 256, h), blk(256);
    hipLaunchKernelGGL(k_large_gradient, g2, blk, 0, s, (const float*)disp, w, h, (float)threshold, flag);
    hipLaunchKernelGGL(k_zero_where, g1, blk, 0, s, disp, n, (const uint8_t*)flag);
    WASS_HIP(c, hipMemsetAsync(best, 0, 8, s));
    hipLaunchKernelGGL(k_uf_init<0>, g1, blk, 0, s, (const void*)disp, n, 0, parent, size);
    hipLaunchKernelGGL(k_uf_merge<0>, g2, blk, 0, s, (const void*)disp, w, h, 0, parent);
    hipLaunchKernelGGL(k_uf_count, g1, blk, 0, s, n, parent, size);
    hipLaunchKernelGGL(k_uf_best, g1, blk, 0, s, n, (const int*)parent, (const unsigned int*)size, best);
    hipLaunchKernelGGL(k_keep_best, g1, blk, 0, s, disp, n, (const int*)parent, (const unsigned long long*)best);
    WASS_HIP(c, hipGetLastError());
    return WASS_OK;
}

int resize_inputs_dev(wass_ctx* c, const uint8_t* src, int w, int h, size_t pitch, uint8_t* dst, int ws, int hs, double fx, double fy,
                      hipStream_t s)
{
    hipLaunchKernelGGL(k_resize_cubic_u8, dim3((ws + 255) / 256, hs), dim3(256), 0, s, src, w, h, pitch, dst, ws, hs, 1.0 / fx, 1.0 / fy);
    WASS_HIP(c, hipGetLastError());
    return WASS_OK;
}

// cv::resize(src, dst, Size(dw, dh), 0, 0, INTER_NEAREST / INTER_CUBIC) of a float map (:903-904)
int resize_f32_dev(wass_ctx* c, const float* src, int sw, int sh, float* dst, int dw, int dh, bool cubic, hipStream_t s)
{
    const double sx = (double)sw / dw, sy = (double)sh / dh;
    if (cubic) hipLaunchKernelGGL(k_resize_f32<1>, dim3((dw + 255) / 256, dh), dim3(256), 0, s, src, sw, sh, dst, dw, dh, sx, sy);
    else hipLaunchKernelGGL(k_resize_f32<0>, dim3((dw + 255) / 256, dh), dim3(256), 0, s, src, sw, sh, dst, dw, dh, sx, sy);
    WASS_HIP(c, hipGetLastError());
    return WASS_OK;
}

}  // namespace wass

using namespace wass;

extern "C" int wass_dense_input_size(int w, int h, double dense_scale, int* ws, int* hs)
{
    if (!ws || !hs || w <= 0 || h <= 0 || !(dense_scale > 0)) return WASS_ERR_INVALID_ARG;
    // cv::resize(src, dst, Size(), fx, fy): dsize = cvRound(size * f); scale > 1 stretches x only (:788-796)
    const double fx = dense_scale, fy = dense_scale > 1.0 ? 1.0 : dense_scale;
    *ws = dense_scale == 1.0 ? w : (int)lrint(w * fx);
    *hs = dense_scale == 1.0 ? h : (int)lrint(h * fy);
    return (*ws > 0 && *hs > 0) ? WASS_OK : WASS_ERR_INVALID_ARG;
}

extern "C" int wass_resize_cubic_u8_dev(wass_ctx* c, const uint8_t* d_src, int sw, int sh, size_t src_stride, uint8_t* d_dst, int dw, int dh)
{
    if (!c || !d_src || !d_dst || sw <= 0 || sh <= 0 || dw <= 0 || dh <= 0 || src_stride < (size_t)sw) return set_err(c, WASS_ERR_INVALID_ARG, "bad argument");
    WASS_HIP(c, hipSetDevice(c->device));
    int rc = wait_uploads(c, d_src, c->stream);
    if (rc) return rc;
    // explicit destination size: the scale factors are the size ratios (cv::resize with fx = fy = 0)
    hipLaunchKernelGGL(k_resize_cubic_u8, dim3((dw + 255) / 256, dh), dim3(256), 0, c->stream, d_src, sw, sh, src_stride, d_dst, dw, dh, (double)sw / dw,
                       (double)sh / dh);
    WASS_HIP(c, hipGetLastError());
    return WASS_OK;
}

extern "C" int wass_biggest_component_by_gradient_dev(wass_ctx* c, float* d_disp, int w, int h, int threshold)
{
    if (!c || !d_disp || w <= 0 || h <= 0 || threshold <= 0) return set_err(c, WASS_ERR_INVALID_ARG, "bad argument");
    WASS_HIP(c, hipSetDevice(c->device));
    int rc = ensure(c, c->tmp_mask, (size_t)w * h);
    if (rc) return rc;
    return biggest_component_dev(c, d_disp, w, h, threshold, (uint8_t*)c->tmp_mask.p, c->ts());
}

// disparity_large_gradient.jpg (wass_stereo.cpp:957-960): the mask the last component extraction of this context used
extern "C" int wass_large_gradient_mask(wass_ctx* c, int w, int h, uint8_t* mask_out)
{
    if (!c || !mask_out || w <= 0 || h <= 0) return set_err(c, WASS_ERR_INVALID_ARG, "bad argument");
    const size_t n = (size_t)w * h;
    if (!c->tmp_mask.p || c->tmp_mask.cap < n) return set_err(c, WASS_ERR_INVALID_ARG, "no component extraction of that size has run on this context");
    WASS_HIP(c, hipSetDevice(c->device));
    WASS_HIP(c, hipMemcpyAsync(mask_out, c->tmp_mask.p, n, hipMemcpyDeviceToHost, c->ts()));
    WASS_HIP(c, hipStreamSynchronize(c->ts()));
    return WASS_OK;
}
