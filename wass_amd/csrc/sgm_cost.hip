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
#include "sgm_step.h"

#include <stdlib.h>

#include <algorithm>
#include <type_traits>

#ifndef WASS_HSUM_XQ
#define WASS_HSUM_XQ 116
#endif
namespace wass {

// ---------------------------------------------------------------------------
// K1: per pixel {v, lo, hi} for the clipped x-Sobel channel and the raw channel
// (lo/hi = min/max over the value and its two half-pixel neighbours).
// ---------------------------------------------------------------------------
// The padded pictures of wass_stereo.cpp:820-831 (zero images of width Wp, the crop copied in at column xo) are never
// materialised: PadImg answers pixel reads of the padded picture from the caller's crop.
struct PadImg {
    const uint8_t* src; size_t pitch; int w, xo;
    __device__ __forceinline__ int at(int X, int y) const
    {
        const int x = X - xo;
        return (unsigned)x < (unsigned)w ? src[(size_t)y * pitch + x] : 0;
    }
};

__device__ __forceinline__ int sobel_at(const PadImg& img, int Wp, int h, int X, int y, int ftzero)
{
    if (X <= 0 || X >= Wp - 1) return ftzero;                 // tab[0]
    const int yn = y > 0 ? y - 1 : 0, ys = y < h - 1 ? y + 1 : y;
    int v = (img.at(X + 1, y) - img.at(X - 1, y)) * 2 + (img.at(X + 1, yn) - img.at(X - 1, yn)) + (img.at(X + 1, ys) - img.at(X - 1, ys));
    v = v < -ftzero ? -ftzero : (v > ftzero ? ftzero : v);
    return v + ftzero;
}
__device__ __forceinline__ int raw_at(const PadImg& img, int Wp, int X, int y, int ftzero)
{
    if (X <= 0 || X >= Wp - 1) return ftzero;                 // tab[0] on the border columns too
    return img.at(X, y);
}

// blockIdx.z = 0: image 1 (the reference image of the match): one 8-byte record per pixel
//   {sobel v, lo, hi, raw v, lo, hi} -- read wave-uniformly by k_hsum_q.
// blockIdx.z = 1: image 2: six u16 planes per row, [y][plane][Wp + pad], stored MIRRORED in x
//   (index Wp-1-X), so that the values a lane needs for consecutive disparities d, d+1, ... at
//   column X (image-2 columns X-d, X-d-1, ...) are consecutive, ascending u16 in memory and arrive
//   as ready-made packed pairs (the same trick OpenCV's calcPixelCostBT uses for its SIMD loop).
// One launch for both pictures; it also clears the frame's status words (flags), which the cost stage ORs into.
__global__ void __launch_bounds__(256) k_prefilter(PadImg img1, PadImg img2, int Wp, int h, int ftzero,
                                                   uint2* __restrict__ out1, unsigned short* __restrict__ out2, int pitch2,
                                                   uint32_t* __restrict__ flags)
{
    __shared__ uint16_t sv[260];                              // (sobel | raw << 8) of columns base-1 .. base+256
    const int base = blockIdx.x * 256;
    const int y = blockIdx.y;
    const bool MIRROR = blockIdx.z != 0;
    if (blockIdx.x == 0 && y == 0 && !MIRROR && threadIdx.x < 16) flags[threadIdx.x] = 0;
    const PadImg& img = MIRROR ? img2 : img1;
    for (int t = threadIdx.x; t < 258; t += 256) {
        const int X = base - 1 + t;
        int s = ftzero, r = ftzero;
        if (X >= 0 && X < Wp) { s = sobel_at(img, Wp, h, X, y, ftzero); r = raw_at(img, Wp, X, y, ftzero); }
        sv[t] = (uint16_t)(s | (r << 8));
    }
    __syncthreads();
    const int X = base + threadIdx.x;
    if (X >= Wp) return;
    const int c = sv[threadIdx.x + 1], l = sv[threadIdx.x], g = sv[threadIdx.x + 2];
    const int s0 = c & 0xff, r0 = c >> 8;
    const int sl = X > 0 ? (s0 + (l & 0xff)) / 2 : s0;
    const int sr = X < Wp - 1 ? (s0 + (g & 0xff)) / 2 : s0;
    const int rl = X > 0 ? (r0 + (l >> 8)) / 2 : r0;
    const int rr = X < Wp - 1 ? (r0 + (g >> 8)) / 2 : r0;
    const int slo = min(min(sl, sr), s0), shi = max(max(sl, sr), s0);
    const int rlo = min(min(rl, rr), r0), rhi = max(max(rl, rr), r0);
    if (!MIRROR) {
        uint2 o;
        o.x = (uint32_t)s0 | ((uint32_t)slo << 8) | ((uint32_t)shi << 16) | ((uint32_t)r0 << 24);
        o.y = (uint32_t)rlo | ((uint32_t)rhi << 8);
        out1[(size_t)y * Wp + X] = o;
    } else {
        const int xm = Wp - 1 - X;
        unsigned short* row = out2 + (size_t)y * 6 * pitch2 + xm;
        row[0] = (unsigned short)s0; row[pitch2] = (unsigned short)slo; row[2 * (size_t)pitch2] = (unsigned short)shi;
        row[3 * (size_t)pitch2] = (unsigned short)r0; row[4 * (size_t)pitch2] = (unsigned short)rlo; row[5 * (size_t)pitch2] = (unsigned short)rhi;
    }
}


// img1 = the picture SGBM takes as its first argument (wass_stereo's right crop, at column D), img2 the other one
// (left crop, at column D + off - comp); w, pitch: the crops'.
int launch_prefilter(wass_ctx* c, const SgmDims& d, const uint8_t* d_img1, const uint8_t* d_img2, size_t pitch)
{
    dim3 grid((d.Wp + 255) / 256, d.h, 2);
    const int pitch2 = bt2_pitch(d.Wp);
    // the slack columns of bt2 are never written; their content only ever feeds padded disparity slots, which k_vsum
    // overwrites with 0xFFFF.
    const PadImg i1 = { d_img1, pitch, d.w, d.D }, i2 = { d_img2, pitch, d.w, d.D + d.off_pos - d.comp };
    KernelClock kc(c);
    kc.begin("k_prefilter", c->stream);
    hipLaunchKernelGGL(k_prefilter, grid, dim3(256), 0, c->stream, i1, i2, d.Wp, d.h, d.ftzero, (uint2*)c->bt1.p,
                       (unsigned short*)c->bt2.p + BT2_FRONT, pitch2, (uint32_t*)c->flags.p);
    kc.end(c->stream);
    WASS_HIP(c, hipGetLastError());
    return WASS_OK;
}

// ---------------------------------------------------------------------------
// K2a: hsum[y][x][d] = sum_{i=-SW2..SW2} pix(y, clamp(x+i, 0, width1-1), d), pix = Birchfield-Tomasi cost of
// the clipped-Sobel channel + (raw channel >> 2).  One wave per (row, chunk of XC columns), lanes = disparities.
//
// Consecutive columns X, X+1, ... read image-2 windows that start one element
// lower each time, so a group of G = 2*NP columns only needs G-1+2*NP consecutive mirrored values per lane.
// They are kept in two register vectors per plane ("lo" = the NP dwords loaded for this group, "hi" = the lo of
// the previous group) and the G shifted windows are cut out with v_alignbit -- one NP-dword load per plane and
// group instead of one per column (4x fewer VMEM instructions and bytes at NP = 2; the per-column version was
// bound by the texture-address path, not by VALU or HBM).  Chunk starts are shifted by `off` so that every group's
// load address is a multiple of G elements.  The window ring lives in wave-private LDS, so any WINSIZE works.
// Groups that touch a replicated border column (first / last chunk of a row) take the per-column path.
// NP dwords with the natural alignment of the group loads (4*NP bytes, capped at 16) -> one dwordx2/x4 load
template <int N> struct __attribute__((aligned(((N & -N) * 4) > 16 ? 16 : ((N & -N) * 4)))) AVec { uint32_t v[N]; };

template <int NP>
__device__ __forceinline__ void bt_eval(const uint2 a1, const us2 (&v)[6][NP], us2 (&out)[NP])
{
    const us2 us = pk_splat(a1.x & 0xff), us0 = pk_splat((a1.x >> 8) & 0xff), us1 = pk_splat((a1.x >> 16) & 0xff);
    const us2 ur = pk_splat(a1.x >> 24), ur0 = pk_splat(a1.y & 0xff), ur1 = pk_splat((a1.y >> 8) & 0xff);
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        const us2 vs = v[0][j], vs0 = v[1][j], vs1 = v[2][j], vr = v[3][j], vr0 = v[4][j], vr1 = v[5][j];
        const us2 cs = pk_min(pk_max(pk_subs(us, vs1), pk_subs(vs0, us)), pk_max(pk_subs(vs, us1), pk_subs(us0, vs)));
        const us2 cr = pk_min(pk_max(pk_subs(ur, vr1), pk_subs(vr0, ur)), pk_max(pk_subs(vr, ur1), pk_subs(ur0, vr)));
        out[j] = cs + (cr >> 2);
    }
}

template <int NP>
__global__ void __launch_bounds__(256) k_hsum_q(const uint2* __restrict__ bt1, const unsigned short* __restrict__ bt2,
                                                int pitch2, int Wp, int width1, int minX1, int minD, int SW2, int XC,
                                                int off, int nchunks, uint32_t* __restrict__ hsum)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t ringbuf[];   // [4 waves][WIN][NP][64]
    constexpr int G = 2 * NP;
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int chunk = blockIdx.x * 4 + wv;
    if (chunk >= nchunks) return;
    const int y = blockIdx.y;
    const int WIN = 2 * SW2 + 1;
    uint32_t* ring = ringbuf + (size_t)wv * WIN * NP * 64 + lane;
    const int xs = chunk * XC + off;                        // first output column of the chunk (may be < 0)
    const int xe = min(xs + XC, width1);
    const int t0 = xs - SW2;                                // first consumed column; (t0 + minX1) has the group phase
    const int steps = xe - xs + 2 * SW2;                    // rounded up to whole groups by the loop below
    const uint2* row1 = bt1 + (size_t)y * Wp + minX1;
    const unsigned short* row2 = bt2 + (size_t)y * 6 * pitch2;

    for (int r = 0; r < WIN * NP; ++r) ring[r * 64] = 0;
    us2 acc[NP];
#pragma unroll
    for (int j = 0; j < NP; ++j) acc[j] = pk_splat(0);

    // mirrored index of the window start at column t0 is e0 = Wp-1-(t0+minX1)+minD == G-1 (mod G)
    int b = (Wp - 1 - (t0 + minX1) + minD) - (G - 1) + G * lane;      // this lane's first element of "lo"
    AVec<NP> hi[6], lo[6], nx[6];
#pragma unroll
    for (int p = 0; p < 6; ++p) {
        hi[p] = *(const AVec<NP>*)(row2 + (size_t)p * pitch2 + (b + G));
        lo[p] = *(const AVec<NP>*)(row2 + (size_t)p * pitch2 + b);
    }
    int slot = 0;
    uint32_t* o = hsum + ((size_t)y * width1 + max(xs, 0)) * (64 * NP) + lane * NP;
    // one finished pixel-cost vector: slide the window, write the column whose window is complete now
    // always: the caller knows (wave-uniformly, once per group) that the column's window is complete and inside the chunk
    auto emit = [&](const us2 (&pix)[NP], int i, int t, auto always) {
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            uint32_t* rs = ring + (slot * NP + j) * 64;
            acc[j] = acc[j] + pix[j] - as_us2(*rs);
            *rs = as_u32(pix[j]);
        }
        slot = slot + 1 == WIN ? 0 : slot + 1;
        const int xo = t - SW2;
        if (decltype(always)::value || (i >= 2 * SW2 && xo >= 0 && xo < xe)) {
            st_stream_vec<NP>(o, acc);
            o += 64 * NP;
        }
    };
    for (int g0 = 0; g0 < steps; g0 += G, b -= G) {
        // the next group's vectors are requested before this group is evaluated (the loads of a group used to be
        // waited for right where they were issued)
#pragma unroll
        for (int p = 0; p < 6; ++p) nx[p] = *(const AVec<NP>*)(row2 + (size_t)p * pitch2 + (b - G));
        const int tg = t0 + g0;
        if (tg >= 0 && tg + G - 1 <= width1 - 1) {                     // wave-uniform: the whole group is inside the row
            auto fast_group = [&](auto always) {
                uint2 a1[G];
#pragma unroll
                for (int s = 0; s < G; ++s) a1[s] = row1[tg + s];          // uniform addresses: one scalar load each
#pragma unroll
                for (int s = 0; s < G; ++s) {
                    const int oo = G - 1 - s;                             // window offset inside [lo | hi], in elements
                    us2 v[6][NP], pix[NP];
#pragma unroll
                    for (int p = 0; p < 6; ++p)
#pragma unroll
                        for (int j = 0; j < NP; ++j) {
                            const int q = oo / 2 + j;                     // dword index into the 2*NP-dword concatenation
                            const uint32_t d0 = q < NP ? lo[p].v[q] : hi[p].v[q - NP];
                            if (oo & 1) {
                                const uint32_t d1 = q + 1 < NP ? lo[p].v[q + 1] : hi[p].v[q + 1 - NP];
                                v[p][j] = as_us2(__builtin_amdgcn_alignbit(d1, d0, 16));
                            } else {
                                v[p][j] = as_us2(d0);
                            }
                        }
                    bt_eval<NP>(a1[s], v, pix);
                    emit(pix, g0 + s, tg + s, always);
                }
            };
            // steady state (every column of the group is written): no per-column tests, no branches around the stores
            if (g0 >= 2 * SW2 && tg - SW2 >= 0 && tg + G - 1 - SW2 < xe) fast_group(std::true_type{});
            else fast_group(std::false_type{});
        } else {                                                       // a replicated border column: per-column loads
            for (int s = 0; s < G; ++s) {
                const int t = tg + s;
                const int tc = t < 0 ? 0 : (t > width1 - 1 ? width1 - 1 : t);
                const unsigned short* src = row2 + (Wp - 1 - (tc + minX1) + minD) + G * lane;   // 2-byte aligned only
                us2 v[6][NP], pix[NP];
#pragma unroll
                for (int p = 0; p < 6; ++p) {
                    AVec<NP> u;
                    __builtin_memcpy(&u, src + (size_t)p * pitch2, sizeof u);
#pragma unroll
                    for (int j = 0; j < NP; ++j) v[p][j] = as_us2(u.v[j]);
                }
                bt_eval<NP>(row1[tc], v, pix);
                emit(pix, g0 + s, t, std::false_type{});
            }
        }
#pragma unroll
        for (int p = 0; p < 6; ++p) { hi[p] = lo[p]; lo[p] = nx[p]; }
    }
}

// Range check and padding of one finished cost vector without branches: padm has 0xFFFF in the half-words of the padded
// disparity slots (d >= D) of this lane, which become the 0xFFFF sentinel; the running maximum of the REAL slots is kept in
// mx and compared with the int16 limit once, at the end of the walk (the per-half-word tests with their exec-mask
// bookkeeping were 40 of the 110 vector instructions per row of k_vsum_col).
template <int NP>
struct RangeCheck {
    uint32_t padm[NP];
    us2 mx[NP];
    __device__ __forceinline__ void init(int dlane, int D)
    {
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            const int d = dlane + 2 * j;
            padm[j] = (d < D ? 0u : 0xFFFFu) | (d + 1 < D ? 0u : 0xFFFF0000u);
            mx[j] = pk_splat(0);
        }
    }
    __device__ __forceinline__ us2 apply(us2 acc, int j)
    {
        mx[j] = pk_max(mx[j], as_us2(as_u32(acc) & ~padm[j]));
        return as_us2(as_u32(acc) | padm[j]);
    }
    __device__ __forceinline__ bool over(us2 lim) const
    {
        bool o = false;
#pragma unroll
        for (int j = 0; j < NP; ++j) o |= (mx[j].x > lim.x) | (mx[j].y > lim.y);
        return o;
    }
};

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
    RangeCheck<NP> rng;
    rng.init(dlane, D);
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
            const us2 v = rng.apply(acc[j], j);
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
    if (__any(rng.over(lim)) && lane == 0) atomicOr(flags, 1u);
}

// ---------------------------------------------------------------------------
// K2b, whole-column form (8-path mode): one wave walks one column from the top row to the bottom row.  That walk
// IS the forward path of the column chain family (path 2), so the wave runs that recurrence on the cost vectors
// it has just produced and stores the family's checkpoints (the state after every K rows) -- the separate
// checkpoint sweep of that family (a full read of C) disappears and its pair kernel can start as soon as C is
// complete.  Rows entering the window are fetched K ahead.
// PATH2 (5-path mode, where path 2 has no partner): the path costs themselves are the first contribution to S and
// are written out (S = L_2), which replaces that path's sweep (a read of C and a read-modify-write of S).
// ---------------------------------------------------------------------------
// SPLIT (8-path pair schedule): the column family is cut in the middle like the row family (half_chain_geometry): wave
// 2x walks rows 0 .. h/2-1 downwards with path 2, wave 2x+1 walks rows h-1 .. h/2 UPWARDS with path 6 -- the vertical
// window sum does not care about the direction.  Both start from the all-zero state at an image border; each leaves its
// final state in endstate[], where the pair kernel of the other half picks it up.  Twice the waves (2 456 columns are
// only 2.4 waves per SIMD) for this kernel and for the family's pair kernel.
// Besides the checkpoints the wave leaves the minimum of the path costs after every step of a checkpointed segment in
// mins[] (K u16 per segment): the pair kernel's forward recomputation reads them back as scalars instead of repeating the
// cross-lane reduction (sgm_step.h, sgm_step_fb).
template <int NP, int K, bool PATH2, bool SPLIT>
__global__ void __launch_bounds__(256) k_vsum_col(const uint32_t* __restrict__ hsum, int width1, int h, int D, int SH2,
                                                  int P1, int P2, uint32_t* __restrict__ C, uint32_t* __restrict__ ckpt,
                                                  uint16_t* __restrict__ mins, int maxseg, uint32_t* __restrict__ S,
                                                  uint32_t* __restrict__ flags, uint32_t* __restrict__ endstate)
{
    static_assert(!SPLIT || !PATH2, "only the pair schedule's column family is split");
    extern __shared__ __attribute__((aligned(16))) uint32_t ringbuf[];   // [4 waves][WIN][NP][64]
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c2 = blockIdx.x * 4 + wv;
    const int x = SPLIT ? c2 >> 1 : c2;
    if (x >= width1) return;
    const bool up = SPLIT && (c2 & 1);
    const int n = SPLIT ? (up ? h - h / 2 : h / 2) : h;              // rows of this walk
    const int ystart = up ? h - 1 : 0, dir = up ? -1 : 1;
    // image row of logical row t of the walk, clamped like the window of the reference (replicated border rows)
    auto yrow = [&](int t) { const int y = ystart + dir * t; return y < 0 ? 0 : (y > h - 1 ? h - 1 : y); };
    const int WIN = 2 * SH2 + 1;
    uint32_t* ring = ringbuf + (size_t)wv * WIN * NP * 64 + lane;
    const size_t vec = 64 * NP, rowstride = (size_t)width1 * vec;
    const uint32_t* hp = hsum + (size_t)x * vec + lane * NP;
    uint32_t* cp = C + (size_t)x * vec + lane * NP;
    uint32_t* ck = ckpt + (size_t)c2 * maxseg * vec + lane * NP;
    uint32_t* mrow = (uint32_t*)(mins + (size_t)c2 * maxseg * K);
    uint32_t ms[K];                                                   // min_d L after every step of the current segment
#pragma unroll
    for (int u = 0; u < K; ++u) ms[u] = 0;
    uint32_t* sp = S + (size_t)x * vec + lane * NP;
    const int dlane = lane * 2 * NP;
    const us2 lim = pk_splat(32767 - P2), P1v = pk_splat(P1), cap = pk_splat(0x7FFF);

    us2 acc[NP];
#pragma unroll
    for (int j = 0; j < NP; ++j) acc[j] = pk_splat(0);
    for (int k = 0; k < WIN; ++k) {                                   // ring slot k holds logical row k - SH2
        const int yy = yrow(k - SH2);
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            const uint32_t v = hp[(size_t)yy * rowstride + j];
            ring[(k * NP + j) * 64] = v;
            acc[j] += as_us2(v);
        }
    }
    RangeCheck<NP> rng;
    rng.init(dlane, D);
    int slot = 0;                                                     // slot of logical row t - SH2: the one leaving next
    PathState<NP> st;
    st.reset();
    const int F = n / K, r = n - F * K;
    // checkpoints k_pair reads: end of segments 0..ncp-1 (a split family keeps all of them, like k_ckpt)
    const int ncp = SPLIT ? F : F - (r > 0 ? 0 : 1);

    // rows entering the window while logical rows tb .. tb+K-1 are finished: logical t + SH2 + 1
    auto fetch = [&](int tb, us2 (&dst)[K][NP]) {
#pragma unroll
        for (int u = 0; u < K; ++u) ld_stream_vec<NP>(hp + (size_t)yrow(tb + u + SH2 + 1) * rowstride, dst[u]);
    };
    auto row = [&](int t, const us2 (&in)[NP], int u) {
        const int y = ystart + dir * t;
        us2 cv[NP], L[NP];
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            const us2 v = rng.apply(acc[j], j);
            cv[j] = v;
            cp[(size_t)y * rowstride + j] = as_u32(v);
        }
        sgm_step<NP>(st, cv, L, P1v, P2);
#pragma unroll
        for (int i = 0; i < K; ++i) ms[i] = i == u ? st.m : ms[i];         // u is a constant wherever row() is inlined
        if (PATH2) {
#pragma unroll
            for (int j = 0; j < NP; ++j) sp[(size_t)y * rowstride + j] = as_u32(pk_min(L[j], cap));
        }
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            uint32_t* rs = ring + (slot * NP + j) * 64;
            acc[j] = acc[j] + in[j] - as_us2(*rs);
            *rs = as_u32(in[j]);
        }
        slot = slot + 1 == WIN ? 0 : slot + 1;
    };

    // K rows at a time with the three dependent pieces decoupled: (a) the K vectors leaving the window are read from
    // the LDS ring back to back, (b) the K cost vectors are finished and stored, (c) the path recurrence runs over
    // them.  Needs K distinct ring slots, i.e. K <= WIN; narrower windows take the row-by-row form.
    auto group = [&](int t0, const us2 (&in)[K][NP]) {
        us2 old[K][NP], cv[K][NP];
#pragma unroll
        for (int u = 0; u < K; ++u) {
            int sl = slot + u;
            sl = sl >= WIN ? sl - WIN : sl;
#pragma unroll
            for (int j = 0; j < NP; ++j) old[u][j] = as_us2(ring[(sl * NP + j) * 64]);
        }
#pragma unroll
        for (int u = 0; u < K; ++u) {
            int sl = slot + u;
            sl = sl >= WIN ? sl - WIN : sl;
#pragma unroll
            for (int j = 0; j < NP; ++j) {
                const us2 v = rng.apply(acc[j], j);
                cv[u][j] = v;
                acc[j] = acc[j] + in[u][j] - old[u][j];
                ring[(sl * NP + j) * 64] = as_u32(in[u][j]);
            }
            st_stream_vec<NP>(cp + (size_t)(ystart + dir * (t0 + u)) * rowstride, cv[u]);
        }
        slot += K;
        slot = slot >= WIN ? slot - WIN : slot;
#pragma unroll
        for (int u = 0; u < K; ++u) {
            us2 L[NP];
            sgm_step<NP>(st, cv[u], L, P1v, P2);
            ms[u] = st.m;
            if (PATH2) {
                us2 s2[NP];
#pragma unroll
                for (int j = 0; j < NP; ++j) s2[j] = pk_min(L[j], cap);
                st_stream_vec<NP>(sp + (size_t)(ystart + dir * (t0 + u)) * rowstride, s2);
            }
        }
    };

    us2 nb[K][NP], nn[K][NP];
    fetch(0, nb);
    const bool batched = K <= WIN;
    for (int s = 0; s < F; ++s) {
        if ((s + 1) * K < n) fetch((s + 1) * K, nn);
        if (batched) {
            group(s * K, nb);
        } else {
#pragma unroll
            for (int u = 0; u < K; ++u) row(s * K + u, nb[u], u);
        }
        if (!PATH2 && s < ncp) {
            st.store_normalised(ck + (size_t)s * vec);
            store_minima<K>(mrow + (size_t)s * (K / 2), ms, lane);
        }
        copy_seg<NP, K>(nb, nn);
    }
#pragma unroll
    for (int u = 0; u < K; ++u)
        if (u < r) row(F * K + u, nb[u], u);
    if (SPLIT) st.store_normalised(endstate + (size_t)c2 * vec + lane * NP);
    if (__any(rng.over(lim)) && lane == 0) atomicOr(flags, 1u);
}

template <int NP>
static int launch_vsum_np(wass_ctx* c, const SgmDims& d, bool plain);

template <int NP>
static int launch_cost_np(wass_ctx* c, const SgmDims& d)
{
    {
        constexpr int G = 2 * NP;
        const int WIN = 2 * d.SW2 + 1;
        // columns per chunk, a multiple of G: 232 where that still leaves >= 16 waves per SIMD-slot round (5 % instead of 10 % of
        // window warm-up: 0.570 against 0.597 ms at config B; 348 is slower again, 60 no faster), 116 on small pictures
        const int XQ = ((size_t)d.h * ((d.width1 + 231) / 232) >= 16384 ? 232 : WASS_HSUM_XQ) / G * G;
        // chunk starts are shifted left by -off in [0, G) so that (t0 + minX1) == (Wp + minD) (mod G)
        const int m = (((d.Wp + d.minD - d.minX1 + d.SW2) % G) + G) % G;
        const int off = m == 0 ? 0 : m - G;
        const int nch = (d.width1 - off + XQ - 1) / XQ;
        const size_t ldsq = (size_t)4 * WIN * NP * 64 * sizeof(uint32_t);
        if (ldsq > 160 * 1024) return set_err(c, WASS_ERR_UNSUPPORTED, "WINSIZE %d too large for the LDS ring", WIN);
        WASS_HIP(c, hipFuncSetAttribute((const void*)k_hsum_q<NP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsq));
        KernelClock kc(c);
        kc.begin("k_hsum_q", c->stream);
        hipLaunchKernelGGL((k_hsum_q<NP>), dim3((nch + 3) / 4, d.h), dim3(256), ldsq, c->stream, (const uint2*)c->bt1.p,
                           (const unsigned short*)c->bt2.p + BT2_FRONT, bt2_pitch(d.Wp), d.Wp, d.width1, d.minX1, d.minD, d.SW2,
                           XQ, off, nch, (uint32_t*)c->hsum.p);
        kc.end(c->stream);
    }
    WASS_HIP(c, hipEventRecord(c->ev[7], c->stream));                        // start of the vertical sum (wass_sgm_timings.vsum_ms)
    return launch_vsum_np<NP>(c, d, false);
}

// The vertical block sum.  plain: the sum alone (k_vsum, row segments), whatever the schedule -- what the stage would cost
// without the column paths riding on it; the roofline accounting times it against the production form (wass_sgm_probe_vsum).
template <int NP>
static int launch_vsum_np(wass_ctx* c, const SgmDims& d, bool plain)
{
    const int YSEG = 128;
    const size_t lds2 = (size_t)4 * (2 * d.SW2 + 1) * NP * 64 * sizeof(uint32_t);
    if (lds2 > 160 * 1024) return set_err(c, WASS_ERR_UNSUPPORTED, "WINSIZE %d too large for the LDS ring", 2 * d.SW2 + 1);
    const CkptLayout lay = ckpt_layout(d);
    KernelClock kc(c);
    if (!plain && (lay.cols_from_cost || lay.path2_from_cost)) {
        constexpr int K = ckpt_k(NP);
        kc.begin(lay.path2_from_cost ? "k_vsum_col(path 2, writes S)" : "k_vsum_col", c->stream);
        int rc = ensure(c, c->ckpt, lay.total);
        if (rc) return rc;
        const dim3 grid((d.width1 + 3) / 4), block(256);
        if (lay.cols_from_cost && lay.split[0]) {
            uint32_t* ckf = (uint32_t*)((char*)c->ckpt.p + lay.off[0]);
            WASS_HIP(c, hipFuncSetAttribute((const void*)k_vsum_col<NP, K, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));
            hipLaunchKernelGGL((k_vsum_col<NP, K, false, true>), dim3((2 * d.width1 + 3) / 4), block, lds2, c->stream, (const uint32_t*)c->hsum.p,
                               d.width1, d.h, d.D, d.SW2, d.P1, d.P2, (uint32_t*)c->C.p, ckf, (uint16_t*)((char*)c->ckpt.p + lay.moff[0]), lay.mseg[0],
                               (uint32_t*)c->S.p, (uint32_t*)c->flags.p, ckf + (size_t)lay.nch[0] * lay.mseg[0] * (64 * NP));
        } else if (lay.cols_from_cost) {
            WASS_HIP(c, hipFuncSetAttribute((const void*)k_vsum_col<NP, K, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));
            hipLaunchKernelGGL((k_vsum_col<NP, K, false, false>), grid, block, lds2, c->stream, (const uint32_t*)c->hsum.p, d.width1, d.h, d.D,
                               d.SW2, d.P1, d.P2, (uint32_t*)c->C.p, (uint32_t*)((char*)c->ckpt.p + lay.off[0]),
                               (uint16_t*)((char*)c->ckpt.p + lay.moff[0]), lay.mseg[0], (uint32_t*)c->S.p, (uint32_t*)c->flags.p,
                               (uint32_t*)nullptr);
        } else {
            WASS_HIP(c, hipFuncSetAttribute((const void*)k_vsum_col<NP, K, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));
            hipLaunchKernelGGL((k_vsum_col<NP, K, true, false>), grid, block, lds2, c->stream, (const uint32_t*)c->hsum.p, d.width1, d.h, d.D,
                               d.SW2, d.P1, d.P2, (uint32_t*)c->C.p, (uint32_t*)c->ckpt.p, (uint16_t*)c->ckpt.p, 0, (uint32_t*)c->S.p,
                               (uint32_t*)c->flags.p, (uint32_t*)nullptr);
        }
        kc.end(c->stream);
        WASS_HIP(c, hipGetLastError());
        return WASS_OK;
    }
    WASS_HIP(c, hipFuncSetAttribute((const void*)k_vsum<NP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));
    dim3 g2((d.width1 + 3) / 4, (d.h + YSEG - 1) / YSEG);
    hipLaunchKernelGGL(k_vsum<NP>, g2, dim3(256), lds2, c->stream, (const uint32_t*)c->hsum.p, d.width1, d.h, d.D,
                       d.SW2, d.P2, YSEG, (uint32_t*)c->C.p, (uint32_t*)c->flags.p);
    WASS_HIP(c, hipGetLastError());
    return WASS_OK;
}

int launch_vsum_only(wass_ctx* c, const SgmDims& d, bool plain)
{
    switch (d.NP) {
        case 1: return launch_vsum_np<1>(c, d, plain);
        case 2: return launch_vsum_np<2>(c, d, plain);
        case 3: return launch_vsum_np<3>(c, d, plain);
        case 4: return launch_vsum_np<4>(c, d, plain);
        case 5: return launch_vsum_np<5>(c, d, plain);
        case 6: return launch_vsum_np<6>(c, d, plain);
        case 7: return launch_vsum_np<7>(c, d, plain);
        case 8: return launch_vsum_np<8>(c, d, plain);
    }
    return set_err(c, WASS_ERR_UNSUPPORTED, "MAX_DISPARITY %d not supported (max 1024)", d.D);
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
