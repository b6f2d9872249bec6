// clahe.hip -- cv::CLAHE::apply for CV_8UC1: the optional contrast equalisation of
// wass_prepare (src/wass_prepare/wass_prepare.cpp:257-262, createCLAHE(CAMx_CLAHE_CLIPLIMIT, Size(T, T)) at :446-449, :471-474).
// Restated as in oracle/clahe_oracle.c (OpenCV 4.5.5 clahe.cpp; parity unpinned): one workgroup per tile builds the clipped
// histogram and its look-up table, one thread per pixel blends the four neighbouring tables.
#include "common.h"

namespace wass {

__device__ __forceinline__ int refl101c(int i, int n) { if (n == 1) return 0; while (i < 0 || i >= n) i = i < 0 ? -i : 2 * n - 2 - i; return i; }

__global__ void __launch_bounds__(256) k_clahe_lut(const uint8_t* __restrict__ src, int w, int h, size_t stride, int tiles_x, int tw, int th,
                                                   int clip, float lut_scale, uint8_t* __restrict__ lut)
{
    __shared__ int hist[256];
    __shared__ int s_clipped;
    const int tx = blockIdx.x % tiles_x, ty = blockIdx.x / tiles_x, t = threadIdx.x;
    hist[t] = 0;
    if (t == 0) s_clipped = 0;
    __syncthreads();
    for (int i = t; i < tw * th; i += 256) {
        const int gx = tx * tw + i % tw, gy = ty * th + i / tw;
        atomicAdd(&hist[src[(size_t)refl101c(gy, h) * stride + refl101c(gx, w)]], 1);
    }
    __syncthreads();
    if (clip > 0) {
        if (hist[t] > clip) { atomicAdd(&s_clipped, hist[t] - clip); hist[t] = clip; }
        __syncthreads();
        const int clipped = s_clipped, batch = clipped / 256;
        int residual = clipped - batch * 256;
        hist[t] += batch;
        if (residual != 0) {
            const int step = max(256 / residual, 1);
            // bins 0, step, 2 step, ... receive one more count each while the remainder lasts (and i < 256)
            if (t % step == 0 && t / step < residual) hist[t] += 1;
        }
        __syncthreads();
    }
    // inclusive prefix sum over the 256 bins (Hillis-Steele in LDS)
    for (int o = 1; o < 256; o <<= 1) {
        const int v = t >= o ? hist[t - o] : 0;
        __syncthreads();
        hist[t] += v;
        __syncthreads();
    }
    const int r = __float2int_rn((float)hist[t] * lut_scale);
    lut[(size_t)blockIdx.x * 256 + t] = (uint8_t)min(max(r, 0), 255);
}

__global__ void __launch_bounds__(256) k_clahe_apply(const uint8_t* __restrict__ src, int w, int h, size_t stride, int tiles_x, int tiles_y,
                                                     float inv_tw, float inv_th, const uint8_t* __restrict__ lut, uint8_t* __restrict__ dst)
{
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= w) return;
    const float tyf = (float)y * inv_th - 0.5f, txf = (float)x * inv_tw - 0.5f;
    int ty1 = (int)floorf(tyf), tx1 = (int)floorf(txf);
    const float ya = tyf - (float)ty1, ya1 = 1.0f - ya, xa = txf - (float)tx1, xa1 = 1.0f - xa;
    int ty2 = min(ty1 + 1, tiles_y - 1), tx2 = min(tx1 + 1, tiles_x - 1);
    ty1 = max(ty1, 0); tx1 = max(tx1, 0);
    const int v = src[(size_t)y * stride + x];
    const float res = ((float)lut[((size_t)ty1 * tiles_x + tx1) * 256 + v] * xa1 + (float)lut[((size_t)ty1 * tiles_x + tx2) * 256 + v] * xa) * ya1 +
                      ((float)lut[((size_t)ty2 * tiles_x + tx1) * 256 + v] * xa1 + (float)lut[((size_t)ty2 * tiles_x + tx2) * 256 + v] * xa) * ya;
    dst[(size_t)y * w + x] = (uint8_t)min(max(__float2int_rn(res), 0), 255);
}

}  // namespace wass

using namespace wass;

extern "C" int wass_clahe_dev(wass_ctx* c, const uint8_t* d_src, int w, int h, size_t src_stride, double clip_limit, int tiles_x, int tiles_y,
                              uint8_t* d_dst)
{
    if (!c || !d_src || !d_dst || w <= 0 || h <= 0 || tiles_x <= 0 || tiles_y <= 0 || src_stride < (size_t)w)
        return set_err(c, WASS_ERR_INVALID_ARG, "bad argument");
    WASS_HIP(c, hipSetDevice(c->device));
    int ew = w, eh = h;
    if (!(w % tiles_x == 0 && h % tiles_y == 0)) { ew = w + (tiles_x - w % tiles_x); eh = h + (tiles_y - h % tiles_y); }
    const int tw = ew / tiles_x, th = eh / tiles_y, area = tw * th;
    int clip = 0;
    if (clip_limit > 0.0) { clip = (int)(clip_limit * area / 256); if (clip < 1) clip = 1; }
    int rc = ensure(c, c->clahe_lut, (size_t)tiles_x * tiles_y * 256);
    if (rc) return rc;
    if ((rc = wait_uploads(c, d_src, c->stream))) return rc;
    uint8_t* lut = (uint8_t*)c->clahe_lut.p;
    hipStream_t s = c->stream;
    hipLaunchKernelGGL(k_clahe_lut, dim3((unsigned)(tiles_x * tiles_y)), dim3(256), 0, s, d_src, w, h, src_stride, tiles_x, tw, th, clip,
                       (float)255 / (float)area, lut);
    hipLaunchKernelGGL(k_clahe_apply, dim3((w + 255) / 256, h), dim3(256), 0, s, d_src, w, h, src_stride, tiles_x, tiles_y, 1.0f / (float)tw,
                       1.0f / (float)th, (const uint8_t*)lut, d_dst);
    WASS_HIP(c, hipGetLastError());
    return WASS_OK;
}

extern "C" int wass_clahe(wass_ctx* c, const uint8_t* src, int w, int h, size_t src_stride, double clip_limit, int tiles_x, int tiles_y,
                          uint8_t* dst)
{
    if (!c || !src || !dst || w <= 0 || h <= 0) return set_err(c, WASS_ERR_INVALID_ARG, "bad argument");
    WASS_HIP(c, hipSetDevice(c->device));
    const size_t n = (size_t)w * h;
    int rc;
    if ((rc = ensure(c, c->tmp_in0, n)) || (rc = ensure(c, c->tmp_in1, n))) return rc;
    WASS_HIP(c, hipMemcpy2DAsync(c->tmp_in0.p, w, src, src_stride, w, h, hipMemcpyHostToDevice, c->stream));
    if ((rc = wass_clahe_dev(c, (const uint8_t*)c->tmp_in0.p, w, h, (size_t)w, clip_limit, tiles_x, tiles_y, (uint8_t*)c->tmp_in1.p))) return rc;
    WASS_HIP(c, hipMemcpyAsync(dst, c->tmp_in1.p, n, hipMemcpyDeviceToHost, c->stream));
    WASS_HIP(c, hipStreamSynchronize(c->stream));
    return WASS_OK;
}
