// grid.hip -- SURVEY.md section 8 row f3: the first step of the gridding stage, straight from the device-resident
// mesh (no mesh_cam.xyzC round trip through the disk).
//
// Reference: gridding/wassgridsurface/wassgridsurface.py:316-365 (_grid_task, algorithm "IDW"):
//   mesh_aligned = (Rpl @ mesh + Tpl, z negated) * CAM_BASELINE          (:318, wass_utils.py:54-61)
//   pts_x = floor((x - xmin) / (xmax - xmin) * (W - 1) + 0.5), pts_y likewise  (:322-326)
//   ZZ = per-cell value of the points that fall into a cell                (:330-345)
//   Zi, mask = IDWInterpolator(KSIZE=5, exp=2.4, reps=1)(ZZ)               (IDWInterpolator.py:23-58)
// Divergence, stated because it cannot be avoided: the reference fills ZZ by ten random sub-samples with a random
// last-writer-wins scatter and takes their nanmedian (:330-345) -- a randomised estimate of the cell's central value with
// no defined result to be identical to.  Here a cell holds the MEAN of all points that fall into it, accumulated in 2^-24
// fixed point so that the result does not depend on the order of the atomics.  Everything after that (the 5x5 inverse-
// distance convolution, the 5x5 morphological closing of the mask) follows IDWInterpolator literally, in fp64.
#include "common.h"

namespace wass {

struct GridDev {
    double R[9], T[3], baseline, xmin, ymin, sx, sy;      // sx = (W - 1) / (xmax - xmin)
    int gw, gh;
};

__global__ void __launch_bounds__(256) k_grid_scatter(const uint8_t* __restrict__ valid, const double* __restrict__ X,
                                                      const double* __restrict__ Y, const double* __restrict__ Z, size_t n, GridDev g,
                                                      long long* __restrict__ sum, unsigned int* __restrict__ cnt)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n || !valid[i]) return;
    const double x = X[i], y = Y[i], z = Z[i];
    const double ax = (g.R[0] * x + g.R[1] * y + g.R[2] * z + g.T[0]) * g.baseline;
    const double ay = (g.R[3] * x + g.R[4] * y + g.R[5] * z + g.T[1]) * g.baseline;
    const double az = -(g.R[6] * x + g.R[7] * y + g.R[8] * z + g.T[2]) * g.baseline;
    const double fx = floor((ax - g.xmin) * g.sx + 0.5), fy = floor((ay - g.ymin) * g.sy + 0.5);
    if (!(fx >= 0 && fx < g.gw && fy >= 0 && fy < g.gh)) return;
    const size_t cidx = (size_t)fy * g.gw + (size_t)fx;
    atomicAdd((unsigned long long*)&sum[cidx], (unsigned long long)(long long)llrint(az * 16777216.0));
    atomicAdd(&cnt[cidx], 1u);
}

// IDWInterpolator.__call__ with reps = 1: I2 = conv(I, K) / (conv(mask, K) + 1e-9); Z = point cells keep their value, empty
// cells take I2; final mask = closing of the point mask with a 5x5 block (dilate, then erode; borders ignored)
__global__ void __launch_bounds__(256) k_grid_idw(const long long* __restrict__ sum, const unsigned int* __restrict__ cnt, int gw, int gh,
                                                  double* __restrict__ zi, uint8_t* __restrict__ dil)
{
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= gw) return;
    const size_t i = (size_t)y * gw + x;
    double num = 0.0, den = 0.0;
    bool any = false;
    for (int dy = -2; dy <= 2; ++dy)
        for (int dx = -2; dx <= 2; ++dx) {
            const int xx = x + dx, yy = y + dy;
            if (xx < 0 || yy < 0 || xx >= gw || yy >= gh) continue;
            const size_t q = (size_t)yy * gw + xx;
            if (!cnt[q]) continue;
            any = true;
            if (!dx && !dy) continue;
            const double k = 1.0 / pow(sqrt((double)(dx * dx + dy * dy)), 2.4);
            num += ((double)sum[q] / 16777216.0 / (double)cnt[q]) * k;
            den += k;
        }
    zi[i] = cnt[i] ? (double)sum[i] / 16777216.0 / (double)cnt[i] : num / (den + 1e-9);
    dil[i] = any;
}
__global__ void __launch_bounds__(256) k_grid_close(const double* __restrict__ zi, const uint8_t* __restrict__ dil, int gw, int gh,
                                                    float* __restrict__ out, uint8_t* __restrict__ mask)
{
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= gw) return;
    bool all = true;
    for (int dy = -2; dy <= 2; ++dy)
        for (int dx = -2; dx <= 2; ++dx) {
            const int xx = x + dx, yy = y + dy;
            if (xx < 0 || yy < 0 || xx >= gw || yy >= gh) continue;      // erosion ignores what lies outside the image
            all = all && dil[(size_t)yy * gw + xx];
        }
    const size_t i = (size_t)y * gw + x;
    out[i] = all ? (float)zi[i] : __builtin_nanf("");
    if (mask) mask[i] = all;
}

}  // namespace wass

using namespace wass;

extern "C" int wass_mesh_grid_idw(wass_ctx* c, const wass_mesh* m, const wass_grid_setup* gs, float* grid_out, uint8_t* mask_out)
{
    if (!c || !m || !gs || !grid_out) return set_err(c, WASS_ERR_INVALID_ARG, "null argument");
    if (gs->width < 2 || gs->height < 2 || !(gs->xmax > gs->xmin) || !(gs->ymax > gs->ymin)) return set_err(c, WASS_ERR_INVALID_ARG, "bad grid");
    WASS_HIP(c, hipSetDevice(c->device));
    const size_t n = m->n(), ng = (size_t)gs->width * gs->height;
    int rc;
    // [sum i64 | zi f64 | out f32 | cnt u32 | dil u8 | mask u8]
    const size_t bytes = ng * (8 + 8 + 4 + 4 + 1 + 1) + 64;
    if ((rc = ensure(c, c->grid, bytes))) return rc;
    long long* sum = (long long*)c->grid.p;
    double* zi = (double*)(sum + ng);
    float* out = (float*)(zi + ng);
    unsigned int* cnt = (unsigned int*)(out + ng);
    uint8_t* dil = (uint8_t*)(cnt + ng);
    uint8_t* mask = dil + ng;
    GridDev g;
    memcpy(g.R, gs->R, sizeof g.R); memcpy(g.T, gs->T, sizeof g.T);
    g.baseline = gs->baseline; g.xmin = gs->xmin; g.ymin = gs->ymin;
    g.sx = (gs->width - 1) / (gs->xmax - gs->xmin); g.sy = (gs->height - 1) / (gs->ymax - gs->ymin);
    g.gw = gs->width; g.gh = gs->height;
    hipStream_t s = c->ts();
    WASS_HIP(c, hipMemsetAsync(sum, 0, ng * 8, s));
    WASS_HIP(c, hipMemsetAsync(cnt, 0, ng * 4, s));
    const dim3 gg((gs->width + 255) / 256, gs->height), blk(256);
    hipLaunchKernelGGL(k_grid_scatter, dim3((unsigned)((n + 255) / 256)), blk, 0, s, m->valid, m->x, m->y, m->z, n, g, sum, cnt);
    hipLaunchKernelGGL(k_grid_idw, gg, blk, 0, s, (const long long*)sum, (const unsigned int*)cnt, g.gw, g.gh, zi, dil);
    hipLaunchKernelGGL(k_grid_close, gg, blk, 0, s, (const double*)zi, (const uint8_t*)dil, g.gw, g.gh, out, mask);
    WASS_HIP(c, hipGetLastError());
    WASS_HIP(c, hipMemcpyAsync(grid_out, out, ng * 4, hipMemcpyDeviceToHost, s));
    if (mask_out) WASS_HIP(c, hipMemcpyAsync(mask_out, mask, ng, hipMemcpyDeviceToHost, s));
    WASS_HIP(c, hipStreamSynchronize(s));
    return WASS_OK;
}
