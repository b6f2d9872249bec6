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

#include <stdlib.h>

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

// MIRROR = false: image 1 (the reference image of the match): one 8-byte record per pixel
//   {sobel v, lo, hi, raw v, lo, hi} -- read wave-uniformly by k_hsum.
// MIRROR = true: image 2: six u16 planes per row, [y][plane][Wp + pad], stored MIRRORED in x
//   (index Wp-1-X), so that the values a lane needs for consecutive disparities d, d+1, ... at
//   column X (image-2 columns X-d, X-d-1, ...) are consecutive, ascending u16 in memory and arrive
//   as ready-made packed pairs (the same trick OpenCV's calcPixelCostBT uses for its SIMD loop).
template <bool MIRROR>
__global__ void __launch_bounds__(256) k_prefilter(const uint8_t* __restrict__ img, int Wp, int h, int ftzero,
                                                   uint2* __restrict__ out1, unsigned short* __restrict__ out2, int pitch2)
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
    if (!MIRROR) {
        uint2 o;
        o.x = (uint32_t)s0 | ((uint32_t)slo << 8) | ((uint32_t)shi << 16) | ((uint32_t)r0 << 24);
        o.y = (uint32_t)rlo | ((uint32_t)rhi << 8);
        out1[(size_t)y * Wp + X] = o;
    } else {
        // every plane is stored BT2_COPIES times, copy k shifted by k elements, so that a lane's run of values can
        // always be fetched with naturally aligned vector loads (copy = start index mod BT2_COPIES; misaligned
        // vector loads are legal on gfx950 but slow)
        const int xm = Wp - 1 - X;
        const unsigned short vals[6] = { (unsigned short)s0, (unsigned short)slo, (unsigned short)shi,
                                         (unsigned short)r0, (unsigned short)rlo, (unsigned short)rhi };
        unsigned short* row = out2 + (size_t)y * (6 * BT2_COPIES) * pitch2;
#pragma unroll
        for (int pl = 0; pl < 6; ++pl)
#pragma unroll
            for (int cpy = 0; cpy < BT2_COPIES; ++cpy)
                if (xm >= cpy) row[(size_t)(BT2_COPIES * pl + cpy) * pitch2 + xm - cpy] = vals[pl];
    }
}


int launch_prefilter(wass_ctx* c, const SgmDims& d)
{
    dim3 grid((d.Wp + 255) / 256, d.h);
    const int pitch2 = bt2_pitch(d.Wp);
    // the slack columns are never written; clear them once per (re)allocation is not enough because sizes change,
    // but their content only ever feeds padded disparity slots, which k_vsum overwrites with 0xFFFF.
    hipLaunchKernelGGL(k_prefilter<false>, grid, dim3(256), 0, c->stream, (const uint8_t*)c->img1.p, d.Wp, d.h,
                       d.ftzero, (uint2*)c->bt1.p, (unsigned short*)nullptr, 0);
    hipLaunchKernelGGL(k_prefilter<true>, grid, dim3(256), 0, c->stream, (const uint8_t*)c->img2.p, d.Wp, d.h,
                       d.ftzero, (uint2*)nullptr, (unsigned short*)c->bt2.p, pitch2);
    WASS_HIP(c, hipGetLastError());
    return WASS_OK;
}

// ---------------------------------------------------------------------------
// K2a: hsum[y][x][d] = sum_{i=-SW2..SW2} pix(y, clamp(x+i, 0, width1-1), d)
// One wave per (row, chunk of XC columns).  Phase 1 computes the BT cost of
// every column of the chunk plus halo into a wave-private LDS strip, phase 2
// slides the window over that strip.  No cross-lane traffic at all.
// ---------------------------------------------------------------------------
template <int N> struct __attribute__((aligned(4))) UVec { uint32_t v[N]; };   // N dwords, at least dword aligned

template <int NP>
__device__ __forceinline__ void bt_cost(const uint2 a1, const unsigned short* __restrict__ m2, int pitch2, int idx,
                                        us2 (&out)[NP])
{
    // image-1 values are uniform over the wave
    const us2 us = pk_splat(a1.x & 0xff), us0 = pk_splat((a1.x >> 8) & 0xff), us1 = pk_splat((a1.x >> 16) & 0xff);
    const us2 ur = pk_splat(a1.x >> 24), ur0 = pk_splat(a1.y & 0xff), ur1 = pk_splat((a1.y >> 8) & 0xff);
    // odd start index -> the copy shifted by one element, at idx-1: the address is always dword aligned
    const unsigned short* p = m2 + (idx & (BT2_COPIES - 1)) * pitch2 + (idx & ~(BT2_COPIES - 1));
    const int pp = BT2_COPIES * pitch2;
    // one NP-dword load per plane (global_load_dwordx2/x3/x4 for NP = 2/3/4)
    const UVec<NP> a0 = *(const UVec<NP>*)(p), a1v = *(const UVec<NP>*)(p + pp), a2 = *(const UVec<NP>*)(p + 2 * pp);
    const UVec<NP> a3 = *(const UVec<NP>*)(p + 3 * pp), a4 = *(const UVec<NP>*)(p + 4 * pp), a5 = *(const UVec<NP>*)(p + 5 * pp);
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        const us2 vs = as_us2(a0.v[j]), vs0 = as_us2(a1v.v[j]), vs1 = as_us2(a2.v[j]);
        const us2 vr = as_us2(a3.v[j]), vr0 = as_us2(a4.v[j]), vr1 = as_us2(a5.v[j]);
        // c0 = max(0, u - v1, v0 - u), c1 = max(0, v - u1, u0 - v), cost = min(c0, c1)
        const us2 cs = pk_min(pk_max(pk_subs(us, vs1), pk_subs(vs0, us)), pk_max(pk_subs(vs, us1), pk_subs(us0, vs)));
        const us2 cr = pk_min(pk_max(pk_subs(ur, vr1), pk_subs(vr0, ur)), pk_max(pk_subs(vr, ur1), pk_subs(ur0, vr)));
        out[j] = cs + (cr >> 2);
    }
}

// LDS strip element: pixel costs are <= 122 + 63 = 185, so an even number of packed pairs is stored as bytes
// (two pairs per dword) -- half the LDS, twice the resident waves.
template <int NP> struct StripFmt { static constexpr bool BYTES = (NP % 2) == 0; static constexpr int DW = BYTES ? NP / 2 : NP; };

template <int NP>
__device__ __forceinline__ void strip_store(uint32_t* __restrict__ strip, int col, int lane, const us2 (&pix)[NP])
{
    if constexpr (StripFmt<NP>::BYTES) {
#pragma unroll
        for (int j = 0; j < NP / 2; ++j)
            strip[(col * (NP / 2) + j) * 64 + lane] = __builtin_amdgcn_perm(as_u32(pix[2 * j + 1]), as_u32(pix[2 * j]), 0x06040200u);
    } else {
#pragma unroll
        for (int j = 0; j < NP; ++j) strip[(col * NP + j) * 64 + lane] = as_u32(pix[j]);
    }
}
template <int NP>
__device__ __forceinline__ void strip_load(const uint32_t* __restrict__ strip, int col, int lane, us2 (&pix)[NP])
{
    if constexpr (StripFmt<NP>::BYTES) {
#pragma unroll
        for (int j = 0; j < NP / 2; ++j) {
            const uint32_t w = strip[(col * (NP / 2) + j) * 64 + lane];
            pix[2 * j] = as_us2(__builtin_amdgcn_perm(0u, w, 0x0c010c00u));
            pix[2 * j + 1] = as_us2(__builtin_amdgcn_perm(0u, w, 0x0c030c02u));
        }
    } else {
#pragma unroll
        for (int j = 0; j < NP; ++j) pix[j] = as_us2(strip[(col * NP + j) * 64 + lane]);
    }
}

template <int NP>
__global__ void __launch_bounds__(64) k_hsum(const uint2* __restrict__ bt1, const unsigned short* __restrict__ bt2,
                                             int pitch2, int Wp, int width1, int minX1, int minD, int SW2, int XC,
                                             uint32_t* __restrict__ hsum)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t strip[];   // [(XC + 2*SW2)][StripFmt::DW][64]
    const int lane = threadIdx.x;
    const int y = blockIdx.y;
    const int xs = blockIdx.x * XC;
    const int xe = min(xs + XC, width1);
    const int n = xe - xs;
    const uint2* row1 = bt1 + (size_t)y * Wp;
    const unsigned short* row2 = bt2 + (size_t)y * (6 * BT2_COPIES) * pitch2;
    const int dbase = minD + lane * 2 * NP;

    // phase 1: four columns per trip so that their (independent) loads are in flight together
    const int cols = n + 2 * SW2;
    for (int i0 = 0; i0 < cols; i0 += 4) {
        us2 pix[4][NP];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            int x = xs - SW2 + min(i0 + q, cols - 1);
            x = x < 0 ? 0 : (x > width1 - 1 ? width1 - 1 : x);
            const int X = x + minX1;
            bt_cost<NP>(row1[X], row2, pitch2, (Wp - 1 - X) + dbase, pix[q]);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (i0 + q < cols) strip_store<NP>(strip, i0 + q, lane, pix[q]);
    }
    // phase 2
    us2 acc[NP], t[NP], u[NP];
#pragma unroll
    for (int j = 0; j < NP; ++j) acc[j] = pk_splat(0);
    for (int i = 0; i < 2 * SW2 + 1; ++i) {
        strip_load<NP>(strip, i, lane, t);
#pragma unroll
        for (int j = 0; j < NP; ++j) acc[j] += t[j];
    }
    uint32_t* o = hsum + ((size_t)y * width1 + xs) * (64 * NP) + lane * NP;
    for (int i = 0; i < n; ++i) {
        if (i > 0) {
            strip_load<NP>(strip, i + 2 * SW2, lane, t);
            strip_load<NP>(strip, i - 1, lane, u);
#pragma unroll
            for (int j = 0; j < NP; ++j) acc[j] = acc[j] + t[j] - u[j];
        }
#pragma unroll
        for (int j = 0; j < NP; ++j) o[(size_t)i * (64 * NP) + j] = as_u32(acc[j]);
    }
}

// Register-ring variant for a compile-time window: the last WIN pixel-cost vectors live in registers (the loop
// is unrolled by WIN so the ring index is static), the running sum is updated as each column is produced, and
// no LDS is touched at all.  Used for the window sizes instantiated below; other sizes take k_hsum.
template <int NP, int WIN>
__global__ void __launch_bounds__(256) k_hsum_ring(const uint2* __restrict__ bt1, const unsigned short* __restrict__ bt2,
                                                   int pitch2, int Wp, int width1, int minX1, int minD, int XC, int nchunks,
                                                   uint32_t* __restrict__ hsum)
{
    constexpr int SW2 = WIN / 2;
    const int lane = threadIdx.x & 63;
    const int chunk = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (chunk >= nchunks) return;
    const int y = blockIdx.y;
    const int xs = chunk * XC;
    const int n = min(XC, width1 - xs);
    const uint2* row1 = bt1 + (size_t)y * Wp;
    const unsigned short* row2 = bt2 + (size_t)y * (6 * BT2_COPIES) * pitch2;
    const int dbase = minD + lane * 2 * NP;
    uint32_t* o = hsum + ((size_t)y * width1 + xs) * (64 * NP) + lane * NP;

    us2 ring[WIN][NP], acc[NP];
#pragma unroll
    for (int r = 0; r < WIN; ++r)
#pragma unroll
        for (int j = 0; j < NP; ++j) ring[r][j] = pk_splat(0);
#pragma unroll
    for (int j = 0; j < NP; ++j) acc[j] = pk_splat(0);
    const int cols = n + 2 * SW2;
    for (int base = 0; base < cols; base += WIN) {
#pragma unroll
        for (int r = 0; r < WIN; ++r) {
            const int i = base + r;
            if (i < cols) {                                                  // wave-uniform
                int x = xs - SW2 + i;
                x = x < 0 ? 0 : (x > width1 - 1 ? width1 - 1 : x);
                const int X = x + minX1;
                us2 pix[NP];
                bt_cost<NP>(row1[X], row2, pitch2, (Wp - 1 - X) + dbase, pix);
#pragma unroll
                for (int j = 0; j < NP; ++j) { acc[j] = acc[j] + pix[j] - ring[r][j]; ring[r][j] = pix[j]; }
                if (i >= 2 * SW2) {
                    uint32_t* oo = o + (size_t)(i - 2 * SW2) * (64 * NP);
#pragma unroll
                    for (int j = 0; j < NP; ++j) oo[j] = as_u32(acc[j]);
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------
// K2b: C[y][x][d] = sum_{j=-SH2..SH2} hsum[clamp(y+j, 0, h-1)][x][d]   (no +P2
// bias is stored; it cancels in the path recurrence and only matters for the
// int16 range check, which is done here).  One wave per (x, segment of rows);
// the 2*SH2+1 rows of the window live in a wave-private LDS ring so that every
// hsum row is fetched once.
// ---------------------------------------------------------------------------
template <int NP>
__global__ void __launch_bounds__(256) k_vsum(const uint32_t* __restrict__ hsum, int width1, int h, int D,
                                              int SH2, int P2, int YSEG, uint32_t* __restrict__ C,
                                              uint32_t* __restrict__ flags)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t ringbuf[];   // [4 waves][WIN][NP][64]
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int x = blockIdx.x * 4 + wv;
    if (x >= width1) return;
    const int WIN = 2 * SH2 + 1;
    uint32_t* ring = ringbuf + (size_t)wv * WIN * NP * 64 + lane;
    const int y0 = blockIdx.y * YSEG, y1 = min(y0 + YSEG, h);
    const size_t rowstride = (size_t)width1 * (64 * NP);
    const uint32_t* hp = hsum + (size_t)x * (64 * NP) + lane * NP;
    uint32_t* cp = C + (size_t)x * (64 * NP) + lane * NP;
    const int dlane = lane * 2 * NP;
    const us2 lim = pk_splat(32767 - P2);

    us2 acc[NP];
#pragma unroll
    for (int j = 0; j < NP; ++j) acc[j] = pk_splat(0);
    for (int k = 0; k < WIN; ++k) {                                   // ring slot k holds row clamp(y0 - SH2 + k)
        int yy = y0 - SH2 + k;
        yy = yy < 0 ? 0 : (yy > h - 1 ? h - 1 : yy);
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            const uint32_t v = hp[(size_t)yy * rowstride + j];
            ring[(k * NP + j) * 64] = v;
            acc[j] += as_us2(v);
        }
    }
    bool over = false;
    int slot = 0;                                                     // slot of row clamp(y - SH2): the one leaving next
    us2 nxt[NP];
    {
        const int ya = min(y0 + SH2 + 1, h - 1);
#pragma unroll
        for (int j = 0; j < NP; ++j) nxt[j] = as_us2(hp[(size_t)ya * rowstride + j]);
    }
    for (int y = y0; y < y1; ++y) {
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            us2 v = acc[j];
            const int d = dlane + 2 * j;
            if (d < D) over |= (v.x > lim.x); else v.x = 0xFFFF;
            if (d + 1 < D) over |= (v.y > lim.y); else v.y = 0xFFFF;
            cp[(size_t)y * rowstride + j] = as_u32(v);
        }
        // slide: row min(y+SH2+1, h-1) enters (already in flight), row clamp(y-SH2) leaves
        us2 cur[NP];
#pragma unroll
        for (int j = 0; j < NP; ++j) cur[j] = nxt[j];
        if (y + 1 < y1) {
            const int ya = min(y + SH2 + 2, h - 1);
#pragma unroll
            for (int j = 0; j < NP; ++j) nxt[j] = as_us2(hp[(size_t)ya * rowstride + j]);
        }
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            acc[j] = acc[j] + cur[j] - as_us2(ring[(slot * NP + j) * 64]);
            ring[(slot * NP + j) * 64] = as_u32(cur[j]);
        }
        slot = slot + 1 == WIN ? 0 : slot + 1;
    }
    if (__any(over) && lane == 0) atomicOr(flags, 1u);
}

template <int NP>
static int launch_cost_np(wass_ctx* c, const SgmDims& d)
{
    const int XC = 48;
    const size_t lds = (size_t)(XC + 2 * d.SW2) * StripFmt<NP>::DW * 64 * sizeof(uint32_t);
    if (lds > 160 * 1024) return set_err(c, WASS_ERR_UNSUPPORTED, "WINSIZE %d too large for the LDS strip", 2 * d.SW2 + 1);
    WASS_HIP(c, hipFuncSetAttribute((const void*)k_hsum<NP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    if (2 * d.SW2 + 1 == 13 && !getenv("WASS_HSUM_LDS")) {
        const int XCR = 13 * 8 - 12, nch = (d.width1 + XCR - 1) / XCR;      // 92 columns + 12 halo = 8 trips of 13
        hipLaunchKernelGGL((k_hsum_ring<NP, 13>), dim3((nch + 3) / 4, d.h), dim3(256), 0, c->stream, (const uint2*)c->bt1.p,
                           (const unsigned short*)c->bt2.p, bt2_pitch(d.Wp), d.Wp, d.width1, d.minX1, d.minD, XCR, nch,
                           (uint32_t*)c->hsum.p);
    } else {
    dim3 g1((d.width1 + XC - 1) / XC, d.h);
    hipLaunchKernelGGL(k_hsum<NP>, g1, dim3(64), lds, c->stream, (const uint2*)c->bt1.p, (const unsigned short*)c->bt2.p,
                       bt2_pitch(d.Wp), d.Wp, d.width1, d.minX1, d.minD, d.SW2, XC, (uint32_t*)c->hsum.p);
    }
    const int YSEG = 128;
    const size_t lds2 = (size_t)4 * (2 * d.SW2 + 1) * NP * 64 * sizeof(uint32_t);
    if (lds2 > 160 * 1024) return set_err(c, WASS_ERR_UNSUPPORTED, "WINSIZE %d too large for the LDS ring", 2 * d.SW2 + 1);
    WASS_HIP(c, hipFuncSetAttribute((const void*)k_vsum<NP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));
    dim3 g2((d.width1 + 3) / 4, (d.h + YSEG - 1) / YSEG);
    hipLaunchKernelGGL(k_vsum<NP>, g2, dim3(256), lds2, c->stream, (const uint32_t*)c->hsum.p, d.width1, d.h, d.D,
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
