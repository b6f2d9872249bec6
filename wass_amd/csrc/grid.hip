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
// no defined result to be identical to.  Two deterministic statistics are offered (wass_mesh_grid_idw_ex):
//   mean    the MEAN of all points that fall into a cell, accumulated in 2^-24 fixed point so that the result does not depend
//           on the order of the atomics (the form of rounds 2-3, and what wass_mesh_grid_idw still computes);
//   median  the exact MEDIAN of the cell's points (round 4): what the reference's estimator converges to, and like it not
//           moved by a few outliers in a cell.  Points are bucketed by cell (count, scan, fill) and every cell sorts its own
//           segment; the result does not depend on the point order either.  Everything after that (the 5x5 inverse-
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

// ---- median mode: points bucketed by cell
// pass 1: the cell of every point (0xFFFFFFFF: not in the grid) and its height, counts per cell
__global__ void __launch_bounds__(256) k_grid_bucket_count(const uint8_t* __restrict__ valid, const double* __restrict__ X,
                                                           const double* __restrict__ Y, const double* __restrict__ Z, size_t n, GridDev g,
                                                           unsigned int* __restrict__ pcell, double* __restrict__ pz, unsigned int* __restrict__ cnt)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    unsigned int cell = 0xFFFFFFFFu;
    double az = 0.0;
    if (valid[i]) {
        const double x = X[i], y = Y[i], z = Z[i];
        const double ax = (g.R[0] * x + g.R[1] * y + g.R[2] * z + g.T[0]) * g.baseline;
        const double ay = (g.R[3] * x + g.R[4] * y + g.R[5] * z + g.T[1]) * g.baseline;
        az = -(g.R[6] * x + g.R[7] * y + g.R[8] * z + g.T[2]) * g.baseline;
        const double fx = floor((ax - g.xmin) * g.sx + 0.5), fy = floor((ay - g.ymin) * g.sy + 0.5);
        if (fx >= 0 && fx < g.gw && fy >= 0 && fy < g.gh) {
            cell = (unsigned int)((size_t)fy * g.gw + (size_t)fx);
            atomicAdd(&cnt[cell], 1u);
        }
    }
    pcell[i] = cell;
    pz[i] = az;
}
// exclusive scan of the cell counts, one workgroup (the grid has at most a few million cells: a few MB)
__global__ void __launch_bounds__(1024) k_grid_scan(const unsigned int* __restrict__ cnt, size_t ng, unsigned int* __restrict__ off)
{
    __shared__ unsigned int part[1024];
    const size_t per = (ng + 1023) / 1024, a = (size_t)threadIdx.x * per, b = a + per < ng ? a + per : ng;
    unsigned int s = 0;
    for (size_t i = a; i < b; ++i) s += cnt[i];
    part[threadIdx.x] = s;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
        const unsigned int v = threadIdx.x >= (unsigned)o ? part[threadIdx.x - o] : 0u;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    unsigned int run = part[threadIdx.x] - s;                // exclusive prefix of this thread's chunk
    for (size_t i = a; i < b; ++i) { off[i] = run; run += cnt[i]; }
}
// pass 2: every point into its cell's segment (the order inside a segment depends on the atomics; the median does not)
__global__ void __launch_bounds__(256) k_grid_bucket_fill(const unsigned int* __restrict__ pcell, const double* __restrict__ pz, size_t n,
                                                          const unsigned int* __restrict__ off, unsigned int* __restrict__ fill,
                                                          double* __restrict__ vals)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const unsigned int c = pcell[i];
    if (c == 0xFFFFFFFFu) return;
    vals[off[c] + atomicAdd(&fill[c], 1u)] = pz[i];
}
// every cell sorts its own segment in place (shell sort: cells hold a handful of points, a few hundred next to the cameras)
// and takes numpy's median: the middle value, or the mean of the two middle values
__global__ void __launch_bounds__(256) k_grid_cell_median(const unsigned int* __restrict__ cnt, const unsigned int* __restrict__ off, size_t ng,
                                                          double* __restrict__ vals, double* __restrict__ cellval)
{
    const size_t c = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (c >= ng) return;
    const unsigned int n = cnt[c];
    if (!n) { cellval[c] = 0.0; return; }
    double* v = vals + off[c];
    for (unsigned int gap = n / 2; gap > 0; gap /= 2)
        for (unsigned int i = gap; i < n; ++i) {
            const double t = v[i];
            unsigned int j = i;
            for (; j >= gap && v[j - gap] > t; j -= gap) v[j] = v[j - gap];
            v[j] = t;
        }
    cellval[c] = (n & 1) ? v[n / 2] : 0.5 * (v[n / 2 - 1] + v[n / 2]);
}
__global__ void __launch_bounds__(256) k_grid_cell_mean(const long long* __restrict__ sum, const unsigned int* __restrict__ cnt, size_t ng,
                                                        double* __restrict__ cellval)
{
    const size_t c = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (c >= ng) return;
    cellval[c] = cnt[c] ? (double)sum[c] / 16777216.0 / (double)cnt[c] : 0.0;
}

// IDWInterpolator.__call__ with reps = 1: I2 = conv(I, K) / (conv(mask, K) + 1e-9); Z = point cells keep their value, empty
// cells take I2; final mask = closing of the point mask with a 5x5 block (dilate, then erode; borders ignored)
__global__ void __launch_bounds__(256) k_grid_idw(const double* __restrict__ cellval, const unsigned int* __restrict__ cnt, int gw, int gh,
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
            num += cellval[q] * k;
            den += k;
        }
    zi[i] = cnt[i] ? cellval[i] : num / (den + 1e-9);
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

extern "C" int wass_mesh_grid_idw_ex(wass_ctx* c, const wass_mesh* m, const wass_grid_setup* gs, int cell_statistic, float* grid_out,
                                     uint8_t* mask_out)
{
    if (!c || !m || !gs || !grid_out) return set_err(c, WASS_ERR_INVALID_ARG, "null argument");
    if (gs->width < 2 || gs->height < 2 || !(gs->xmax > gs->xmin) || !(gs->ymax > gs->ymin)) return set_err(c, WASS_ERR_INVALID_ARG, "bad grid");
    if (cell_statistic != WASS_GRID_CELL_MEAN && cell_statistic != WASS_GRID_CELL_MEDIAN) return set_err(c, WASS_ERR_INVALID_ARG, "unknown cell statistic");
    WASS_HIP(c, hipSetDevice(c->device));
    const size_t n = m->n(), ng = (size_t)gs->width * gs->height;
    if (ng > 0x7FFFFFF0ull) return set_err(c, WASS_ERR_UNSUPPORTED, "grid too large");
    const bool median = cell_statistic == WASS_GRID_CELL_MEDIAN;
    int rc;
    // [sum i64 | zi f64 | cellval f64 | out f32 | cnt u32 | off u32 | fill u32 | dil u8 | mask u8]  (+ per point: pz f64, vals f64, pcell u32)
    const size_t bytes = ng * (8 + 8 + 8 + 4 + 4 + 4 + 4 + 1 + 1) + (median ? n * (8 + 8 + 4) : 0) + 256;
    if ((rc = ensure(c, c->grid, bytes))) return rc;
    long long* sum = (long long*)c->grid.p;
    double* zi = (double*)(sum + ng);
    double* cellval = zi + ng;
    double* pz = cellval + ng;                               // median mode only
    double* vals = pz + (median ? n : 0);
    float* out = (float*)(vals + (median ? n : 0));
    unsigned int* cnt = (unsigned int*)(out + ng);
    unsigned int* off = cnt + ng;
    unsigned int* fill = off + ng;
    unsigned int* pcell = fill + ng;                         // median mode only
    uint8_t* dil = (uint8_t*)(pcell + (median ? n : 0));
    uint8_t* mask = dil + ng;
    GridDev g;
    memcpy(g.R, gs->R, sizeof g.R); memcpy(g.T, gs->T, sizeof g.T);
    g.baseline = gs->baseline; g.xmin = gs->xmin; g.ymin = gs->ymin;
    g.sx = (gs->width - 1) / (gs->xmax - gs->xmin); g.sy = (gs->height - 1) / (gs->ymax - gs->ymin);
    g.gw = gs->width; g.gh = gs->height;
    hipStream_t s = c->ts();
    const dim3 gg((gs->width + 255) / 256, gs->height), blk(256), gp((unsigned)((n + 255) / 256)), gc((unsigned)((ng + 255) / 256));
    WASS_HIP(c, hipMemsetAsync(cnt, 0, ng * 4, s));
    if (median) {
        WASS_HIP(c, hipMemsetAsync(fill, 0, ng * 4, s));
        hipLaunchKernelGGL(k_grid_bucket_count, gp, blk, 0, s, m->valid, m->x, m->y, m->z, n, g, pcell, pz, cnt);
        hipLaunchKernelGGL(k_grid_scan, dim3(1), dim3(1024), 0, s, (const unsigned int*)cnt, ng, off);
        hipLaunchKernelGGL(k_grid_bucket_fill, gp, blk, 0, s, (const unsigned int*)pcell, (const double*)pz, n, (const unsigned int*)off, fill, vals);
        hipLaunchKernelGGL(k_grid_cell_median, gc, blk, 0, s, (const unsigned int*)cnt, (const unsigned int*)off, ng, vals, cellval);
    } else {
        WASS_HIP(c, hipMemsetAsync(sum, 0, ng * 8, s));
        hipLaunchKernelGGL(k_grid_scatter, gp, blk, 0, s, m->valid, m->x, m->y, m->z, n, g, sum, cnt);
        hipLaunchKernelGGL(k_grid_cell_mean, gc, blk, 0, s, (const long long*)sum, (const unsigned int*)cnt, ng, cellval);
    }
    hipLaunchKernelGGL(k_grid_idw, gg, blk, 0, s, (const double*)cellval, (const unsigned int*)cnt, g.gw, g.gh, zi, dil);
    hipLaunchKernelGGL(k_grid_close, gg, blk, 0, s, (const double*)zi, (const uint8_t*)dil, g.gw, g.gh, out, mask);
    WASS_HIP(c, hipGetLastError());
    WASS_HIP(c, hipMemcpyAsync(grid_out, out, ng * 4, hipMemcpyDeviceToHost, s));
    if (mask_out) WASS_HIP(c, hipMemcpyAsync(mask_out, mask, ng, hipMemcpyDeviceToHost, s));
    WASS_HIP(c, hipStreamSynchronize(s));
    return WASS_OK;
}

extern "C" int wass_mesh_grid_idw(wass_ctx* c, const wass_mesh* m, const wass_grid_setup* gs, float* grid_out, uint8_t* mask_out)
{
    return wass_mesh_grid_idw_ex(c, m, gs, WASS_GRID_CELL_MEAN, grid_out, mask_out);
}
