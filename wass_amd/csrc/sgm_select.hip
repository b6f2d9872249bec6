// sgm_select.hip -- K4b (right-view disparity + left-right check; the per-pixel winner-take-all is fused into
// the last aggregation kernel, sgm_step.h: wta_batch) and K5 (3x3 median of the int16 map + crop).
//
// Replaces the per-row tail of OpenCV's computeDisparitySGBM and the
// medianBlur in StereoSGBMImpl::compute, reached
// from wass_stereo/wass_stereo.cpp:837, and the colRange crop of :839.
#include "common.h"

namespace wass {

// K4b: one workgroup per image row: right-view disparity by an LDS atomic
// minimum (equal costs: the larger x wins, as in the descending-x loop of the
// original), then the left-right consistency check.
__global__ void __launch_bounds__(256) k_lrcheck(const int16_t* __restrict__ sel_d16,
                                                 const uint32_t* __restrict__ sel_key, int width1, int Wp,
                                                 int minX1, int minD, int d12, int16_t* __restrict__ raw)
{
    extern __shared__ uint32_t d2key[];             // [Wp]
    const int y = blockIdx.x;
    const int INVALID = (minD - 1) * 16;
    for (int X = threadIdx.x; X < Wp; X += blockDim.x) d2key[X] = 0xFFFFFFFFu;
    __syncthreads();
    for (int x = threadIdx.x; x < width1; x += blockDim.x) {
        const uint32_t k = sel_key[(size_t)y * width1 + x];
        if (k != 0xFFFFFFFFu) {
            const int minS = (int)(k >> 16);
            const int d = (int)(int16_t)(k & 0xFFFF);
            const int X2 = x + minX1 - d - minD;
            // disp2cost starts at MAX_COST and is replaced only by a strictly smaller cost
            if (minS < 32767 && X2 >= 0 && X2 < Wp)
                atomicMin(&d2key[X2], ((uint32_t)minS << 16) | (uint32_t)(width1 - 1 - x));
        }
    }
    __syncthreads();
    int16_t* out = raw + (size_t)y * Wp;
    for (int X = threadIdx.x; X < Wp; X += blockDim.x) {
        int d1 = INVALID;
        if (X >= minX1) {
            d1 = sel_d16[(size_t)y * width1 + (X - minX1)];
            if (d1 != INVALID) {
                const int dlo = d1 >> 4, dhi = (d1 + 15) >> 4;
                const int xlo = X - dlo, xhi = X - dhi;
                bool c1 = false, c2 = false;
                if (0 <= xlo && xlo < Wp) {
                    const uint32_t k = d2key[xlo];
                    const int disp2 = k == 0xFFFFFFFFu ? INVALID : (width1 - 1 - (int)(k & 0xFFFF)) + minX1 - xlo;
                    c1 = disp2 >= minD && abs(disp2 - dlo) > d12;
                }
                if (0 <= xhi && xhi < Wp) {
                    const uint32_t k = d2key[xhi];
                    const int disp2 = k == 0xFFFFFFFFu ? INVALID : (width1 - 1 - (int)(k & 0xFFFF)) + minX1 - xhi;
                    c2 = disp2 >= minD && abs(disp2 - dhi) > d12;
                }
                if (c1 && c2) d1 = INVALID;
            }
        }
        out[X] = (int16_t)d1;
    }
}

int launch_select(wass_ctx* c, const SgmDims& d)
{
    hipLaunchKernelGGL(k_lrcheck, dim3(d.h), dim3(256), (size_t)d.Wp * sizeof(uint32_t), c->stream,
                       (const int16_t*)c->sel_d16.p, (const uint32_t*)c->sel_key.p, d.width1, d.Wp, d.minX1,
                       d.minD, d.d12, (int16_t*)c->raw.p);
    WASS_HIP(c, hipGetLastError());
    return WASS_OK;
}

// K5: cv::medianBlur(disp, disp, 3) on CV_16S (replicate border) over the padded
// map, writing only the columns [D, D+w) that wass_stereo keeps (:839).
__device__ __forceinline__ void cswap(int& a, int& b) { const int lo = min(a, b), hi = max(a, b); a = lo; b = hi; }

__global__ void __launch_bounds__(256) k_median_crop(const int16_t* __restrict__ raw, int Wp, int h, int col0,
                                                     int w, int16_t* __restrict__ out)
{
    const int j = blockIdx.x * 256 + threadIdx.x;
    const int y = blockIdx.y;
    if (j >= w) return;
    const int X = col0 + j;
    const int xs[3] = { max(X - 1, 0), X, min(X + 1, Wp - 1) };
    const int ys[3] = { max(y - 1, 0), y, min(y + 1, h - 1) };
    int v[9];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) v[a * 3 + b] = raw[(size_t)ys[a] * Wp + xs[b]];
    // 19-exchange median-of-9 network
    cswap(v[1], v[2]); cswap(v[4], v[5]); cswap(v[7], v[8]);
    cswap(v[0], v[1]); cswap(v[3], v[4]); cswap(v[6], v[7]);
    cswap(v[1], v[2]); cswap(v[4], v[5]); cswap(v[7], v[8]);
    cswap(v[0], v[3]); cswap(v[5], v[8]); cswap(v[4], v[7]);
    cswap(v[3], v[6]); cswap(v[1], v[4]); cswap(v[2], v[5]);
    cswap(v[4], v[7]); cswap(v[4], v[2]); cswap(v[6], v[4]);
    cswap(v[4], v[2]);
    out[(size_t)y * w + j] = (int16_t)v[4];
}

int launch_median_crop(wass_ctx* c, const SgmDims& d, int16_t* d_out)
{
    dim3 grid((d.w + 255) / 256, d.h);
    hipLaunchKernelGGL(k_median_crop, grid, dim3(256), 0, c->stream, (const int16_t*)c->raw.p, d.Wp, d.h, d.D,
                       d.w, d_out);
    WASS_HIP(c, hipGetLastError());
    return WASS_OK;
}

int launch_median_full(wass_ctx* c, const SgmDims& d, int16_t* d_padded_out)
{
    dim3 grid((d.Wp + 255) / 256, d.h);
    hipLaunchKernelGGL(k_median_crop, grid, dim3(256), 0, c->stream, (const int16_t*)c->raw.p, d.Wp, d.h, 0, d.Wp, d_padded_out);
    WASS_HIP(c, hipGetLastError());
    return WASS_OK;
}

}  // namespace wass
