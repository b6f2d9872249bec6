// sgm_cost.hip -- K1 (prefilter + Birchfield-Tomasi intervals) and K2 (block-
// summed matching cost volume C) for gfx950.
//
// Replaces OpenCV's calcPixelCostBT and the hsum / C sliding sums inside
// computeDisparitySGBM, which the reference
// reaches through dense_stereo->compute (wass_stereo/wass_stereo.cpp:837).
//
// Layout: disparities are the innermost dimension of every volume and map
// onto the 64 lanes of a wavefront, NP packed u16 pairs per lane
// (d = 2*NP*lane + j).  One wave therefore reads/writes one contiguous
// 256*NP-byte vector per pixel -- fully coalesced -- and all arithmetic runs
// on v_pk_*_u16.  Slots d >= D hold 0xFFFF in C and never win a minimum.
#include "common.h"

namespace wass {

// ---------------------------------------------------------------------------
// K1: per pixel {v, lo, hi} for the clipped x-Sobel channel and the raw channel
// (lo/hi = min/max over the value and its two half-pixel neighbours).
// ---------------------------------------------------------------------------
__device__ __forceinline__ int sobel_at(const uint8_t* img, int Wp, int h, int X, int y, int ftzero)
{
    if (X <= 0 || X >= Wp - 1) return ftzero;                 // tab[0]
    const int yn = y > 0 ? y - 1 : 0, ys = y < h - 1 ? y + 1 : y;
    const uint8_t* r = img + (size_t)y * Wp;
    const uint8_t* rn = img + (size_t)yn * Wp;
    const uint8_t* rs = img + (size_t)ys * Wp;
    int v = (r[X + 1] - r[X - 1]) * 2 + (rn[X + 1] - rn[X - 1]) + (rs[X + 1] - rs[X - 1]);
    v = v < -ftzero ? -ftzero : (v > ftzero ? ftzero : v);
    return v + ftzero;
}
__device__ __forceinline__ int raw_at(const uint8_t* img, int Wp, int X, int y, int ftzero)
{
    if (X <= 0 || X >= Wp - 1) return ftzero;                 // tab[0] on the border columns too
    return img[(size_t)y * Wp + X];
}

__global__ void __launch_bounds__(256) k_prefilter(const uint8_t* __restrict__ img, int Wp, int h,
                                                   int ftzero, uint2* __restrict__ out)
{
    const int X = blockIdx.x * 256 + threadIdx.x;
    const int y = blockIdx.y;
    if (X >= Wp) return;
    int s0 = sobel_at(img, Wp, h, X, y, ftzero);
    int r0 = raw_at(img, Wp, X, y, ftzero);
    int sl = X > 0 ? (s0 + sobel_at(img, Wp, h, X - 1, y, ftzero)) / 2 : s0;
    int sr = X < Wp - 1 ? (s0 + sobel_at(img, Wp, h, X + 1, y, ftzero)) / 2 : s0;
    int rl = X > 0 ? (r0 + raw_at(img, Wp, X - 1, y, ftzero)) / 2 : r0;
    int rr = X < Wp - 1 ? (r0 + raw_at(img, Wp, X + 1, y, ftzero)) / 2 : r0;
    int slo = min(min(sl, sr), s0), shi = max(max(sl, sr), s0);
    int rlo = min(min(rl, rr), r0), rhi = max(max(rl, rr), r0);
    uint2 o;
    o.x = (uint32_t)s0 | ((uint32_t)slo << 8) | ((uint32_t)shi << 16) | ((uint32_t)r0 << 24);
    o.y = (uint32_t)rlo | ((uint32_t)rhi << 8);
    out[(size_t)y * Wp + X] = o;
}

int launch_prefilter(wass_ctx* c, const SgmDims& d)
{
    dim3 grid((d.Wp + 255) / 256, d.h);
    hipLaunchKernelGGL(k_prefilter, grid, dim3(256), 0, c->stream, (const uint8_t*)c->img1.p, d.Wp, d.h,
                       d.ftzero, (uint2*)c->bt1.p);
    hipLaunchKernelGGL(k_prefilter, grid, dim3(256), 0, c->stream, (const uint8_t*)c->img2.p, d.Wp, d.h,
                       d.ftzero, (uint2*)c->bt2.p);
    WASS_HIP(c, hipGetLastError());
    return WASS_OK;
}

// ---------------------------------------------------------------------------
// K2a: hsum[y][x][d] = sum_{i=-SW2..SW2} pix(y, clamp(x+i, 0, width1-1), d)
// One wave per (row, chunk of XC columns).  Phase 1 computes the BT cost of
// every column of the chunk plus halo into a wave-private LDS strip, phase 2
// slides the window over that strip.  No cross-lane traffic at all.
// ---------------------------------------------------------------------------
template <int NP>
__device__ __forceinline__ void bt_cost(const uint2 a1, const uint2* __restrict__ row2, int X, int dbase,
                                        int Wp, us2 (&out)[NP])
{
    // image-1 values are uniform over the wave
    const us2 us = pk_splat(a1.x & 0xff), us0 = pk_splat((a1.x >> 8) & 0xff), us1 = pk_splat((a1.x >> 16) & 0xff);
    const us2 ur = pk_splat(a1.x >> 24), ur0 = pk_splat(a1.y & 0xff), ur1 = pk_splat((a1.y >> 8) & 0xff);
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        // disparities dbase+2j and dbase+2j+1 -> image-2 columns X-d (descending)
        int xa = X - (dbase + 2 * j), xb = xa - 1;
        xa = xa < 0 ? 0 : xa;             // only padded slots (d >= D) can fall off the row
        xb = xb < 0 ? 0 : xb;
        const uint2 va = row2[xa], vb = row2[xb];
        us2 vs, vs0, vs1, vr, vr0, vr1;
        vs.x = va.x & 0xff; vs.y = vb.x & 0xff;
        vs0.x = (va.x >> 8) & 0xff; vs0.y = (vb.x >> 8) & 0xff;
        vs1.x = (va.x >> 16) & 0xff; vs1.y = (vb.x >> 16) & 0xff;
        vr.x = va.x >> 24; vr.y = vb.x >> 24;
        vr0.x = va.y & 0xff; vr0.y = vb.y & 0xff;
        vr1.x = (va.y >> 8) & 0xff; vr1.y = (vb.y >> 8) & 0xff;
        // c0 = max(0, u - v1, v0 - u), c1 = max(0, v - u1, u0 - v), cost = min(c0, c1)
        us2 cs = pk_min(pk_max(pk_subs(us, vs1), pk_subs(vs0, us)), pk_max(pk_subs(vs, us1), pk_subs(us0, vs)));
        us2 cr = pk_min(pk_max(pk_subs(ur, vr1), pk_subs(vr0, ur)), pk_max(pk_subs(vr, ur1), pk_subs(ur0, vr)));
        out[j] = cs + (cr >> 2);
    }
}

template <int NP>
__global__ void __launch_bounds__(64) k_hsum(const uint2* __restrict__ bt1, const uint2* __restrict__ bt2,
                                             int Wp, int width1, int minX1, int minD, int SW2, int XC,
                                             uint32_t* __restrict__ hsum)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t strip[];   // [(XC + 2*SW2)][NP][64]
    const int lane = threadIdx.x;
    const int y = blockIdx.y;
    const int xs = blockIdx.x * XC;
    const int xe = min(xs + XC, width1);
    const int n = xe - xs;
    const uint2* row1 = bt1 + (size_t)y * Wp;
    const uint2* row2 = bt2 + (size_t)y * Wp;
    const int dbase = minD + lane * 2 * NP;

    // phase 1
    for (int i = 0; i < n + 2 * SW2; ++i) {
        int x = xs - SW2 + i;
        x = x < 0 ? 0 : (x > width1 - 1 ? width1 - 1 : x);
        const int X = x + minX1;
        us2 pix[NP];
        bt_cost<NP>(row1[X], row2, X, dbase, Wp, pix);
#pragma unroll
        for (int j = 0; j < NP; ++j) strip[(i * NP + j) * 64 + lane] = as_u32(pix[j]);
    }
    // phase 2
    us2 acc[NP];
#pragma unroll
    for (int j = 0; j < NP; ++j) acc[j] = pk_splat(0);
    for (int i = 0; i < 2 * SW2 + 1; ++i)
#pragma unroll
        for (int j = 0; j < NP; ++j) acc[j] += as_us2(strip[(i * NP + j) * 64 + lane]);
    uint32_t* o = hsum + ((size_t)y * width1 + xs) * (64 * NP) + lane * NP;
    for (int i = 0; i < n; ++i) {
        if (i > 0) {
#pragma unroll
            for (int j = 0; j < NP; ++j)
                acc[j] = acc[j] + as_us2(strip[((i + 2 * SW2) * NP + j) * 64 + lane]) -
                         as_us2(strip[((i - 1) * NP + j) * 64 + lane]);
        }
#pragma unroll
        for (int j = 0; j < NP; ++j) o[(size_t)i * (64 * NP) + j] = as_u32(acc[j]);
    }
}

// ---------------------------------------------------------------------------
// K2b: C[y][x][d] = sum_{j=-SH2..SH2} hsum[clamp(y+j, 0, h-1)][x][d]   (no +P2
// bias is stored; it cancels in the path recurrence and only matters for the
// int16 range check, which is done here).  One wave per (x, segment of rows).
// ---------------------------------------------------------------------------
template <int NP>
__global__ void __launch_bounds__(256) k_vsum(const uint32_t* __restrict__ hsum, int width1, int h, int D,
                                              int SH2, int P2, int YSEG, uint32_t* __restrict__ C,
                                              uint32_t* __restrict__ flags)
{
    const int lane = threadIdx.x & 63;
    const int x = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (x >= width1) return;
    const int y0 = blockIdx.y * YSEG, y1 = min(y0 + YSEG, h);
    const size_t rowstride = (size_t)width1 * (64 * NP);
    const uint32_t* hp = hsum + (size_t)x * (64 * NP) + lane * NP;
    uint32_t* cp = C + (size_t)x * (64 * NP) + lane * NP;
    const int dlane = lane * 2 * NP;
    const us2 lim = pk_splat(32767 - P2);

    us2 acc[NP];
#pragma unroll
    for (int j = 0; j < NP; ++j) acc[j] = pk_splat(0);
    for (int k = -SH2; k <= SH2; ++k) {
        int yy = y0 + k;
        yy = yy < 0 ? 0 : (yy > h - 1 ? h - 1 : yy);
#pragma unroll
        for (int j = 0; j < NP; ++j) acc[j] += as_us2(hp[(size_t)yy * rowstride + j]);
    }
    bool over = false;
    for (int y = y0; y < y1; ++y) {
        if (y > y0) {
            const int ya = min(y + SH2, h - 1), ys = max(y - SH2 - 1, 0);
#pragma unroll
            for (int j = 0; j < NP; ++j)
                acc[j] = acc[j] + as_us2(hp[(size_t)ya * rowstride + j]) - as_us2(hp[(size_t)ys * rowstride + j]);
        }
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            us2 v = acc[j];
            const int d = dlane + 2 * j;
            if (d < D) over |= (v.x > lim.x); else v.x = 0xFFFF;
            if (d + 1 < D) over |= (v.y > lim.y); else v.y = 0xFFFF;
            cp[(size_t)y * rowstride + j] = as_u32(v);
        }
    }
    if (__any(over) && lane == 0) atomicOr(flags, 1u);
}

template <int NP>
static int launch_cost_np(wass_ctx* c, const SgmDims& d)
{
    const int XC = 48;
    const size_t lds = (size_t)(XC + 2 * d.SW2) * NP * 64 * sizeof(uint32_t);
    if (lds > 160 * 1024) return set_err(c, WASS_ERR_UNSUPPORTED, "WINSIZE %d too large for the LDS strip", 2 * d.SW2 + 1);
    WASS_HIP(c, hipFuncSetAttribute((const void*)k_hsum<NP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    dim3 g1((d.width1 + XC - 1) / XC, d.h);
    hipLaunchKernelGGL(k_hsum<NP>, g1, dim3(64), lds, c->stream, (const uint2*)c->bt1.p, (const uint2*)c->bt2.p,
                       d.Wp, d.width1, d.minX1, d.minD, d.SW2, XC, (uint32_t*)c->hsum.p);
    const int YSEG = 64;
    dim3 g2((d.width1 + 3) / 4, (d.h + YSEG - 1) / YSEG);
    hipLaunchKernelGGL(k_vsum<NP>, g2, dim3(256), 0, c->stream, (const uint32_t*)c->hsum.p, d.width1, d.h, d.D,
                       d.SW2, d.P2, YSEG, (uint32_t*)c->C.p, (uint32_t*)c->flags.p);
    WASS_HIP(c, hipGetLastError());
    return WASS_OK;
}

int launch_cost_volume(wass_ctx* c, const SgmDims& d)
{
    switch (d.NP) {
        case 1: return launch_cost_np<1>(c, d);
        case 2: return launch_cost_np<2>(c, d);
        case 3: return launch_cost_np<3>(c, d);
        case 4: return launch_cost_np<4>(c, d);
        case 5: return launch_cost_np<5>(c, d);
        case 6: return launch_cost_np<6>(c, d);
        case 7: return launch_cost_np<7>(c, d);
        case 8: return launch_cost_np<8>(c, d);
    }
    return set_err(c, WASS_ERR_UNSUPPORTED, "MAX_DISPARITY %d not supported (max 1024)", d.D);
}

}  // namespace wass
