// sgm_tile.hip -- K3, tile-fused schedule: path aggregation with S kept on the chip.
//
// Same recurrences as sgm_aggregate.hip (reference call site
// wass_stereo/wass_stereo.cpp:837), different decomposition -- see tile_geom.h:
//
//   1. k_edge_sweep, once per path: one wave per chain walks the whole chain reading C and stores nothing but the
//      (normalised) state with which the path enters each T x T tile.  Seven pure read streams (the forward column
//      path is swept by the cost stage itself, k_vsum_col), all independent, all in flight at once.
//   2. k_tile, one workgroup per tile: the tile's cost vectors are read once into LDS; family by family (rows, columns,
//      diagonals, anti-diagonals) wave w rebuilds its share of the path costs from the entry states and adds them
//      into an S tile that lives in LDS; the last family finishes S in registers and runs the winner-take-all.
//
// HBM traffic per cell (8 paths, T = 12): 7 x 2 B (sweeps) + 2 B (tile) + ~3.9 B (edge states written and read)
// = ~20 B, against 28 B for the pair schedule, whose partial S made three round trips through HBM.  S itself is only
// written in debug mode (wass_ctx_set_debug), for the parity tests.
#include "sgm_quad.h"
#include "tile_geom.h"

#include <stdlib.h>

#ifndef WASS_UQ2
#define WASS_UQ2 8            // prefetch depth (steps) of the quad sweep at NP = 2
#endif

namespace wass {

struct EdgePtrs {
    uint32_t* row[4][2];      // [family][0 forward / 1 backward]
    uint32_t* col[4][2];
};

// ---------------------------------------------------------------------------
// One path, every chain: keep the states that enter a tile.  (dx,dy) = direction of travel.
// ---------------------------------------------------------------------------
template <int NP, int T, int U>
__global__ void __launch_bounds__(256) k_edge_sweep(const uint32_t* __restrict__ C, uint32_t* __restrict__ rowedge,
                                                    uint32_t* __restrict__ coledge, int width1, int h, int dx, int dy,
                                                    int P1, int P2, int nchains)
{
    const int lane = threadIdx.x & 63;
    const int c = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform
    if (c >= nchains) return;
    int x0, y0, n;
    chain_geometry(c, dx, dy, width1, h, x0, y0, n);
    const long long vec = 64 * NP;
    const long long step = ((long long)dy * width1 + dx) * vec;
    const uint32_t* cp = C + ((long long)y0 * width1 + x0) * vec + lane * NP;
    uint32_t* re = rowedge + lane * NP;
    uint32_t* ce = coledge + lane * NP;
    const us2 P1v = pk_splat(P1);
    PathState<NP> st;
    st.reset();
    int x = x0, y = y0;
    auto advance = [&](const us2 (&cv)[NP]) {
        us2 L[NP];
        sgm_step<NP>(st, cv, L, P1v, P2);
        long long idx = 0;
        const int k = edge_store_slot(x, y, dx, dy, T, width1, h, idx);          // wave-uniform
        if (k == 1) st.store_normalised(re + idx * vec);
        else if (k == 2) st.store_normalised(ce + idx * vec);
        x += dx; y += dy;
    };
    const int F = n / U, r = n - F * U;
    us2 cb[U][NP], cn[U][NP];
    if (F > 0) load_seg<NP, U, false>(cp, step, U, cb);
    else load_seg<NP, U, true>(cp, step, r, cb);
    for (int g = 0; g < F; ++g) {
        if (g + 1 < F) load_seg<NP, U, false>(cp + U * step, step, U, cn);
        else if (r > 0) load_seg<NP, U, true>(cp + U * step, step, r, cn);
#pragma unroll
        for (int u = 0; u < U; ++u) advance(cb[u]);
        copy_seg<NP, U>(cb, cn);
        cp += U * step;
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
        if (u < r) advance(cb[u]);
}

// ---------------------------------------------------------------------------
// The same sweep in quad layout (sgm_quad.h): four chains per wave, one per DPP row.  The chains of a wave are T apart
// (c, c + T, c + 2T, c + 3T) and all start on the same image border, so they cross tile borders in the same steps: the
// crossing test is scalar and a wave stores edge states in 2 of every T steps.  The rows of a wave have different lengths
// only by a few steps (diagonals); the common part of the walk runs without any guard, K steps of loads ahead.
// ---------------------------------------------------------------------------
__host__ __device__ inline int quad_group_waves(int chains, int T) { return (chains + 4 * T - 1) / (4 * T) * T; }

template <int NP, int T, int U>
__global__ void __launch_bounds__(256) k_edge_sweep_q(const uint32_t* __restrict__ C, uint32_t* __restrict__ rowedge,
                                                      uint32_t* __restrict__ coledge, int width1, int h, int dx, int dy,
                                                      int P1, int P2)
{
    constexpr int NQ = 4 * NP;
    const int lane = threadIdx.x & 63, r = lane >> 4, s = lane & 15;
    const int wv = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform
    // two groups of chains: those starting on the top / bottom image row, then those starting on the left / right column
    const int Gt = dy != 0 ? width1 : 0, Gs = dy == 0 ? h : (dx == 0 ? 0 : h - 1);
    const int Wt = quad_group_waves(Gt, T);
    const bool top = wv < Wt;
    const int wl = top ? wv : wv - Wt, G = top ? Gt : Gs;
    const int i0 = (wl / T) * 4 * T + wl % T;                         // index of row 0's chain inside its group
    if (i0 >= G) return;
    const bool active = i0 + r * T < G;
    const int ii = active ? i0 + r * T : i0;
    int x0, y0, n;
    if (top) {
        x0 = ii; y0 = dy > 0 ? 0 : h - 1;
        n = dx == 0 ? h : min(dx > 0 ? width1 - x0 : x0 + 1, h);
    } else {
        const int kk = dy == 0 ? ii : ii + 1;
        x0 = dx > 0 ? 0 : width1 - 1;
        y0 = dy >= 0 ? kk : h - 1 - kk;
        n = dy == 0 ? width1 : min(width1, dy > 0 ? h - y0 : y0 + 1);
    }
    if (!active) n = 0;
    int nmin = 1 << 30, nmax = 0;                                     // common part / longest of the four walks
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int nq = __builtin_amdgcn_readlane(n, 16 * q);
        nmax = max(nmax, nq);
        if (nq > 0) nmin = min(nmin, nq);
    }
    const long long vec = 64 * NP;
    const long long step = ((long long)dy * width1 + dx) * vec;
    const long long stepr = active ? step : 0;                        // an idle row re-reads the first pixel of row 0's chain
    const uint32_t* cp = C + ((long long)y0 * width1 + x0) * vec + s * NQ;
    uint32_t* re = rowedge + s * NQ;
    uint32_t* ce = coledge + s * NQ;
    const us2 P1v = pk_splat(P1), P2v = pk_splat(P2);
    QState<NQ> st;
    st.reset();
    // steps until the walks leave their tile column / tile row (the same for every row of the wave): scalar counters
    const int X = __builtin_amdgcn_readfirstlane(x0), Y = __builtin_amdgcn_readfirstlane(y0);
    int cx = dx > 0 ? T - 1 - X % T : (dx < 0 ? X % T : (1 << 29));
    int cy = dy > 0 ? T - 1 - Y % T : (dy < 0 ? Y % T : (1 << 29));
    int k = 0;                                                        // step about to be processed (wave-uniform)
    auto advance = [&](const us2 (&cv)[NQ]) {
        us2 L[NQ];
        qstep<NQ>(st, cv, L, P1v, P2v);
        --cx; --cy;
        if ((cx | cy) < 0) {                                          // the pixels just finished were the last ones inside a tile
            const bool crow = cy < 0, ccol = cx < 0;
            us2 nv[NQ];
            st.normalised(nv);
            if (k + 1 < n) {
                const int nx = x0 + (k + 1) * dx, ny = y0 + (k + 1) * dy;
                if (crow) q_st<NQ>(re + ((long long)(ny / T) * width1 + nx) * vec, nv);
                else q_st<NQ>(ce + ((long long)(nx / T) * h + ny) * vec, nv);
            }
            if (crow) cy += T;
            if (ccol) cx += T;
        }
        ++k;
    };
    const int F = nmin / U;
    us2 bufA[U][NQ], bufB[U][NQ];
    const uint32_t* lp = cp;                                          // next pixel to be requested
    auto request = [&](us2 (&buf)[U][NQ]) {
#pragma unroll
        for (int u = 0; u < U; ++u) q_ld<NQ>(lp + u * stepr, buf[u]);
        lp += U * stepr;
    };
    if (F > 0) request(bufA);
    int g = 0;
    for (; g + 1 < F; g += 2) {                                       // two groups per iteration: no register copies
        request(bufB);
#pragma unroll
        for (int u = 0; u < U; ++u) advance(bufA[u]);
        if (g + 2 < F) request(bufA);
#pragma unroll
        for (int u = 0; u < U; ++u) advance(bufB[u]);
    }
    if (g < F) {
#pragma unroll
        for (int u = 0; u < U; ++u) advance(bufA[u]);
    }
    // the last nmin % U common steps and the steps only some rows still have: one at a time, loads guarded per row
    while (k < nmax) {
        us2 cv[NQ];
#pragma unroll
        for (int j = 0; j < NQ; ++j) cv[j] = pk_splat(0);
        if (k < n) q_ld<NQ>(lp, cv);
        lp += stepr;
        advance(cv);
    }
}

// ---------------------------------------------------------------------------
// k_tile
// ---------------------------------------------------------------------------
template <int NP>
__device__ __forceinline__ void load_entry(int kind, long long idx, const uint32_t* __restrict__ rowp,
                                           const uint32_t* __restrict__ colp, int lane, us2 (&v)[NP])
{
    if (kind == 1) ld_stream_vec<NP>(rowp + idx * (64 * NP) + lane * NP, v);
    else if (kind == 2) ld_stream_vec<NP>(colp + idx * (64 * NP) + lane * NP, v);
    else {
#pragma unroll
        for (int j = 0; j < NP; ++j) v[j] = pk_splat(0);
    }
}

template <int NP>
__device__ __forceinline__ void lds_ld(const uint32_t* p, us2 (&v)[NP])
{
#pragma unroll
    for (int j = 0; j < NP; ++j) v[j] = as_us2(p[j]);
}
template <int NP>
__device__ __forceinline__ void lds_st(uint32_t* p, const us2 (&v)[NP])
{
#pragma unroll
    for (int j = 0; j < NP; ++j) p[j] = as_u32(v[j]);
}

// One workgroup per tile, one wave per tile row; the four families are the iterations of ONE rolled loop so that the
// unrolled recurrence exists once in the instruction stream (the kernel has to stay well inside the 64 KB instruction
// cache that two CUs share: a version with one inlined copy per family was 70 KB and ran 4x slower).
//   smode 0: S = L (first family)   1: S += L   2: last family -- S is finished in registers and handed to the
//   winner-take-all.   both: the family's backward path is aggregated too (always for rows; the others only in MODE_HH).
// Full runs (T cells) take the register path: cost vectors and partial sums of the T cells live in registers, forward
// and backward recurrence advance together.  Clipped runs of partial tiles (right / bottom image border, ~1 % of the
// tiles) take a rolled cell-by-cell path through LDS.
template <int NP, int T, bool HH>
__global__ void __launch_bounds__(T * 64) k_tile(const uint32_t* __restrict__ C, const EdgePtrs ep, int width1, int h,
                                                 int ntx, int P1, int P2, int D, int minD, int uniq, int keepS,
                                                 uint32_t* __restrict__ Sg, int16_t* __restrict__ sel_d16,
                                                 uint32_t* __restrict__ sel_key, int fx, int fy)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t tile_lds[];   // C tile, then S tile: [T*T][64*NP] each
    constexpr int vec = 64 * NP;
    uint32_t* Ct = tile_lds;
    uint32_t* St = tile_lds + T * T * vec;
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);       // wave-uniform
    // fx < 0: every tile of the image.  Otherwise only the tiles outside the fx x fy block of complete tiles that
    // k_tile_q handles: the partial (or complete) tile columns on the right, then the tile rows at the bottom.
    int tx, ty;
    if (fx < 0) { tx = blockIdx.x % ntx; ty = blockIdx.x / ntx; }
    else {
        const int nty = (h + T - 1) / T, ncr = ntx - fx, nright = ncr * nty;
        int b = blockIdx.x;
        if (b < nright) { tx = fx + b % ncr; ty = b / ncr; }
        else { b -= nright; tx = b % fx; ty = fy + b / fx; }
    }
    const int X0 = tx * T, Y0 = ty * T;
    const int tw = min(T, width1 - X0), th = min(T, h - Y0);
    const us2 P1v = pk_splat(P1), cap = pk_splat(0x7FFF);

    // cost vectors of tile row w: HBM -> registers -> LDS (one contiguous run of tw vectors)
    if (w < th) {
        const uint32_t* cp = C + ((long long)(Y0 + w) * width1 + X0) * vec + lane * NP;
        us2 row[T][NP];
#pragma unroll
        for (int i = 0; i < T; ++i)
            if (i < tw) ld_stream_vec<NP>(cp + (long long)i * vec, row[i]);
#pragma unroll
        for (int i = 0; i < T; ++i)
            if (i < tw) lds_st<NP>(Ct + (w * T + i) * vec + lane * NP, row[i]);
    }
    // the row phase of wave w reads exactly the cells wave w has just written: no barrier needed before it
#pragma unroll 1
    for (int fam = FAM_ROWS; fam <= FAM_ANTI; ++fam) {
        const int smode = fam == FAM_ROWS ? 0 : (fam == FAM_ANTI ? 2 : 1);
        const bool both = HH || fam == FAM_ROWS;
        TileSeg a, b;
        tile_segments(fam, w, T, tw, th, a, b);
        if (a.n == 0) { a = b; b.n = 0; }
        const int n1 = a.n, ntot = a.n + b.n;                     // wave-uniform
        int fdx, fdy;
        family_dir(fam, fdx, fdy);
        const int stride = fdy * T + fdx;                         // LDS cells per step along the run
        const int base_a = a.sy * T + a.sx, base_b = b.sy * T + b.sx - n1 * stride;
        auto cell = [&](int i) { return (i < n1 ? base_a : base_b) + i * stride; };
        auto pixel = [&](int cl) { return (long long)(Y0 + cl / T) * width1 + (X0 + cl % T); };

        if (ntot > 0) {
            // states with which the forward path enters run a / run b and the backward path enters them from the other end
            us2 eFA[NP], eFB[NP], eBA[NP], eBB[NP];
            {
                long long idx = 0;
                int k = edge_entry_slot(X0 + a.sx, Y0 + a.sy, fdx, fdy, T, width1, h, idx);
                load_entry<NP>(k, idx, ep.row[fam][0], ep.col[fam][0], lane, eFA);
                k = b.n ? edge_entry_slot(X0 + b.sx, Y0 + b.sy, fdx, fdy, T, width1, h, idx) : 0;
                load_entry<NP>(k, idx, ep.row[fam][0], ep.col[fam][0], lane, eFB);
                if (both) {
                    k = edge_entry_slot(X0 + a.sx + (a.n - 1) * fdx, Y0 + a.sy + (a.n - 1) * fdy, -fdx, -fdy, T, width1, h, idx);
                    load_entry<NP>(k, idx, ep.row[fam][1], ep.col[fam][1], lane, eBA);
                    k = b.n ? edge_entry_slot(X0 + b.sx + (b.n - 1) * fdx, Y0 + b.sy + (b.n - 1) * fdy, -fdx, -fdy, T, width1, h, idx) : 0;
                    load_entry<NP>(k, idx, ep.row[fam][1], ep.col[fam][1], lane, eBB);
                }
            }
            PathState<NP> fw, bw;
            fw.load_normalised(eFA);
            if (b.n) bw.load_normalised(eBB); else bw.load_normalised(eBA);
            if (ntot == T) {
                us2 cv[T][NP], acc[T][NP];
#pragma unroll
                for (int i = 0; i < T; ++i) {
                    lds_ld<NP>(Ct + cell(i) * vec + lane * NP, cv[i]);
#pragma unroll
                    for (int j = 0; j < NP; ++j) acc[i][j] = pk_splat(0);
                }
                if (both) {
                    // forward over slots 0..T-1 and backward over slots T-1..0, advanced together statement by statement
#pragma unroll
                    for (int i = 0; i < T; ++i) {
                        const int v = T - 1 - i;
                        if (i == n1) fw.load_normalised(eFB);             // run b starts here
                        if (v == n1 - 1 && b.n) bw.load_normalised(eBA);  // the backward path leaves run b, enters run a
                        us2 Lf[NP], Lb[NP];
                        sgm_step_pair<NP>(fw, cv[i], Lf, bw, cv[v], Lb, P1v, P2);
#pragma unroll
                        for (int j = 0; j < NP; ++j) {
                            acc[i][j] = pk_adds(acc[i][j], Lf[j]);
                            acc[v][j] = pk_adds(acc[v][j], Lb[j]);
                        }
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < T; ++i) {
                        if (i == n1) fw.load_normalised(eFB);
                        sgm_step<NP>(fw, cv[i], acc[i], P1v, P2);
                    }
                }
                // fold into the S tile
#pragma unroll
                for (int i = 0; i < T; ++i) {
                    uint32_t* sp = St + cell(i) * vec + lane * NP;
                    if (smode != 0) {
                        us2 s[NP];
                        lds_ld<NP>(sp, s);
#pragma unroll
                        for (int j = 0; j < NP; ++j) acc[i][j] = pk_adds(acc[i][j], s[j]);
                    }
#pragma unroll
                    for (int j = 0; j < NP; ++j) acc[i][j] = pk_min(acc[i][j], cap);
                    if (smode != 2) lds_st<NP>(sp, acc[i]);
                    else if (keepS) st_stream_vec<NP>(Sg + pixel(cell(i)) * vec + lane * NP, acc[i]);
                }
                if (smode == 2) {
                    int res_d;
                    uint32_t res_k;
                    wta_batch_eval<NP, T>(acc, lane, D, minD, uniq, res_d, res_k);
                    if (lane < T) {                                   // lane u holds the result of slot u
                        const long long px = pixel((lane < n1 ? base_a : base_b) + lane * stride);
                        sel_d16[px] = (int16_t)res_d;
                        sel_key[px] = res_k;
                    }
                }
            } else {
                // clipped runs: cell by cell through LDS (S += forward, then S += backward, then the selection)
#pragma unroll 1
                for (int i = 0; i < ntot; ++i) {
                    if (i == n1) fw.load_normalised(eFB);
                    us2 cvv[NP], L[NP], s[NP];
                    lds_ld<NP>(Ct + cell(i) * vec + lane * NP, cvv);
                    sgm_step<NP>(fw, cvv, L, P1v, P2);
                    uint32_t* sp = St + cell(i) * vec + lane * NP;
                    if (smode != 0) {
                        lds_ld<NP>(sp, s);
#pragma unroll
                        for (int j = 0; j < NP; ++j) L[j] = pk_adds(L[j], s[j]);
                    }
#pragma unroll
                    for (int j = 0; j < NP; ++j) L[j] = pk_min(L[j], cap);
                    lds_st<NP>(sp, L);
                }
                if (both) {
#pragma unroll 1
                    for (int v = ntot - 1; v >= 0; --v) {
                        if (v == n1 - 1 && b.n) bw.load_normalised(eBA);
                        us2 cvv[NP], L[NP], s[NP];
                        lds_ld<NP>(Ct + cell(v) * vec + lane * NP, cvv);
                        sgm_step<NP>(bw, cvv, L, P1v, P2);
                        uint32_t* sp = St + cell(v) * vec + lane * NP;
                        lds_ld<NP>(sp, s);
#pragma unroll
                        for (int j = 0; j < NP; ++j) L[j] = pk_min(pk_adds(L[j], s[j]), cap);
                        lds_st<NP>(sp, L);
                    }
                }
                if (smode == 2) {
#pragma unroll 1
                    for (int i = 0; i < ntot; ++i) {
                        us2 s[NP];
                        lds_ld<NP>(St + cell(i) * vec + lane * NP, s);
                        const long long px = pixel(cell(i));
                        if (keepS) st_stream_vec<NP>(Sg + px * vec + lane * NP, s);
                        wta_select<NP>(s, lane, D, minD, uniq, sel_d16 + px, sel_key + px);
                    }
                }
            }
        }
        if (fam != FAM_ANTI) __syncthreads();
    }
}

// ---------------------------------------------------------------------------
// k_tile_q: the tile kernel in quad layout (sgm_quad.h) for COMPLETE tiles.  T/4 waves per tile; DPP row r of wave w
// plays the part of "wave 4w + r" of k_tile (tile row / column / pair of diagonals 4w + r), so all the per-run geometry
// is per-lane data and the four runs of a wave advance in lock-step.  The cost vectors and partial sums of a wave's
// 4 x T cells stay in registers during a phase (one wave per SIMD: the register file is not the constraint, the
// instruction count is).  LDS holds a vector as [chunk q/4][lane s][4 dwords], so that the 16 lanes of a row touch 256
// contiguous bytes per ds_read_b128 / ds_write_b128.  The winner-take-all runs after the last family on the S tile in
// LDS, in the one-pixel-per-wave layout of sgm_step.h (wta_batch_eval).
// ---------------------------------------------------------------------------
template <int NQ>
__device__ __forceinline__ void qlds_ld(const uint32_t* p, us2 (&v)[NQ])          // p: cell base + 4 * s
{
#pragma unroll
    for (int j = 0; j < NQ; j += 4) {
        const wass_u32x4 x = *(const wass_u32x4*)(p + (j / 4) * 64);
        v[j] = as_us2(x.x); v[j + 1] = as_us2(x.y); v[j + 2] = as_us2(x.z); v[j + 3] = as_us2(x.w);
    }
}
template <int NQ>
__device__ __forceinline__ void qlds_st(uint32_t* p, const us2 (&v)[NQ])
{
#pragma unroll
    for (int j = 0; j < NQ; j += 4) {
        wass_u32x4 x = { as_u32(v[j]), as_u32(v[j + 1]), as_u32(v[j + 2]), as_u32(v[j + 3]) };
        *(wass_u32x4*)(p + (j / 4) * 64) = x;
    }
}

template <int NQ>
__device__ __forceinline__ void q_entry(int kind, long long idx, const uint32_t* __restrict__ rowp,
                                        const uint32_t* __restrict__ colp, int s, us2 (&v)[NQ])
{
#pragma unroll
    for (int j = 0; j < NQ; ++j) v[j] = pk_splat(0);
    if (kind == 1) q_ld<NQ>(rowp + idx * (16 * NQ) + s * NQ, v);
    else if (kind == 2) q_ld<NQ>(colp + idx * (16 * NQ) + s * NQ, v);
}

template <int NP, int T, bool HH>
__global__ void __launch_bounds__(T / 4 * 64) k_tile_q(const uint32_t* __restrict__ C, const EdgePtrs ep, int width1, int h,
                                                       int fx, int P1, int P2, int D, int minD, int uniq, int keepS,
                                                       uint32_t* __restrict__ Sg, int16_t* __restrict__ sel_d16,
                                                       uint32_t* __restrict__ sel_key)
{
    static_assert(T % 4 == 0, "one DPP row per tile row");
    constexpr int NQ = 4 * NP, vec = 64 * NP, NW = T / 4;
    extern __shared__ __attribute__((aligned(16))) uint32_t tile_lds[];   // C tile, then S tile: [T*T][vec] each
    uint32_t* Ct = tile_lds;
    uint32_t* St = tile_lds + T * T * vec;
    const int lane = threadIdx.x & 63, r = lane >> 4, s = lane & 15;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);       // wave-uniform
    const int wv = 4 * w + r;                                             // the run(s) this DPP row works on
    const int tx = blockIdx.x % fx, ty = blockIdx.x / fx;
    const int X0 = tx * T, Y0 = ty * T;
    const us2 P1v = pk_splat(P1), P2v = pk_splat(P2), cap = pk_splat(0x7FFF);

    {   // cost vectors of tile row wv: HBM -> registers -> LDS
        const uint32_t* cp = C + ((long long)(Y0 + wv) * width1 + X0) * vec + s * NQ;
        us2 row[T][NQ];
#pragma unroll
        for (int i = 0; i < T; ++i) q_ld<NQ>(cp + (long long)i * vec, row[i]);
#pragma unroll
        for (int i = 0; i < T; ++i) qlds_st<NQ>(Ct + (wv * T + i) * vec + 4 * s, row[i]);
    }
    // the row phase of a DPP row reads exactly the cells it has just written: no barrier needed before it
#pragma unroll 1
    for (int fam = FAM_ROWS; fam <= FAM_ANTI; ++fam) {
        const int smode = fam == FAM_ROWS ? 0 : (fam == FAM_ANTI ? 2 : 1);
        const bool both = HH || fam == FAM_ROWS;
        int fdx, fdy;
        family_dir(fam, fdx, fdy);
        // runs a (n1 cells) and b (T - n1 cells) of a complete tile, tile_segments() without the clipping
        int asx, asy, bsx, bsy, n1;
        if (fam == FAM_ROWS) { asx = 0; asy = wv; n1 = T; bsx = bsy = 0; }
        else if (fam == FAM_COLS) { asx = wv; asy = 0; n1 = T; bsx = bsy = 0; }
        else if (fam == FAM_DIAG) { asx = 0; asy = T - 1 - wv; n1 = wv + 1; bsx = wv + 1; bsy = 0; }
        else { asx = wv; asy = 0; n1 = wv + 1; bsx = T - 1; bsy = wv + 1; }
        const bool hasb = n1 < T;
        const int stride = fdy * T + fdx;
        const int base_a = asy * T + asx, base_b = bsy * T + bsx - n1 * stride;
        auto cell = [&](int i) { return (i < n1 ? base_a : base_b) + i * stride; };

        us2 eFA[NQ], eFB[NQ], eBA[NQ], eBB[NQ];
        {
            long long idx = 0;
            int k = edge_entry_slot(X0 + asx, Y0 + asy, fdx, fdy, T, width1, h, idx);
            q_entry<NQ>(k, idx, ep.row[fam][0], ep.col[fam][0], s, eFA);
            k = hasb ? edge_entry_slot(X0 + bsx, Y0 + bsy, fdx, fdy, T, width1, h, idx) : 0;
            q_entry<NQ>(k, idx, ep.row[fam][0], ep.col[fam][0], s, eFB);
            if (both) {
                k = edge_entry_slot(X0 + asx + (n1 - 1) * fdx, Y0 + asy + (n1 - 1) * fdy, -fdx, -fdy, T, width1, h, idx);
                q_entry<NQ>(k, idx, ep.row[fam][1], ep.col[fam][1], s, eBA);
                k = hasb ? edge_entry_slot(X0 + bsx + (T - n1 - 1) * fdx, Y0 + bsy + (T - n1 - 1) * fdy, -fdx, -fdy, T, width1, h, idx) : 0;
                q_entry<NQ>(k, idx, ep.row[fam][1], ep.col[fam][1], s, eBB);
            }
        }
        us2 cv[T][NQ], acc[T][NQ];
#pragma unroll
        for (int i = 0; i < T; ++i) qlds_ld<NQ>(Ct + cell(i) * vec + 4 * s, cv[i]);
        QState<NQ> fw, bw;
        fw.load_normalised(eFA);
        bw.load_normalised(eBA);
        if (hasb) bw.load_normalised(eBB);
        const bool diagonal = fam >= FAM_DIAG;                            // only these have a second run
#pragma unroll
        for (int i = 0; i < T; ++i) {
            const int v = T - 1 - i;
            if (diagonal) {
                if (__builtin_amdgcn_ballot_w64(i == n1) != 0) {          // run b starts here for some DPP row
                    const bool sel = i == n1;
#pragma unroll
                    for (int j = 0; j < NQ; ++j) fw.L[j] = sel ? eFB[j] : fw.L[j];
                    fw.pm = sel ? 0u : fw.pm;
                }
                if (both && __builtin_amdgcn_ballot_w64(hasb && v == n1 - 1) != 0) {   // backward path: run b -> run a
                    const bool sel = hasb && v == n1 - 1;
#pragma unroll
                    for (int j = 0; j < NQ; ++j) bw.L[j] = sel ? eBA[j] : bw.L[j];
                    bw.pm = sel ? 0u : bw.pm;
                }
            }
            us2 Lf[NQ], Lb[NQ];
            qstep<NQ>(fw, cv[i], Lf, P1v, P2v);
            if (both) qstep<NQ>(bw, cv[v], Lb, P1v, P2v);
            // slot i meets the forward path now and the backward path in iteration T - 1 - i: the first visitor assigns
            if (!both || i < T - 1 - i) {
#pragma unroll
                for (int j = 0; j < NQ; ++j) acc[i][j] = Lf[j];
            } else {
#pragma unroll
                for (int j = 0; j < NQ; ++j) acc[i][j] = pk_adds(acc[i][j], Lf[j]);
            }
            if (both) {
                if (v > i) {
#pragma unroll
                    for (int j = 0; j < NQ; ++j) acc[v][j] = Lb[j];
                } else {
#pragma unroll
                    for (int j = 0; j < NQ; ++j) acc[v][j] = pk_adds(acc[v][j], Lb[j]);
                }
            }
        }
        // fold into the S tile (the last family leaves the finished S there for the selection pass)
#pragma unroll
        for (int i = 0; i < T; ++i) {
            const int cl = cell(i);
            uint32_t* sp = St + cl * vec + 4 * s;
            if (smode != 0) {
                us2 sv[NQ];
                qlds_ld<NQ>(sp, sv);
#pragma unroll
                for (int j = 0; j < NQ; ++j) acc[i][j] = pk_adds(acc[i][j], sv[j]);
            }
#pragma unroll
            for (int j = 0; j < NQ; ++j) acc[i][j] = pk_min(acc[i][j], cap);
            qlds_st<NQ>(sp, acc[i]);
            if (smode == 2 && keepS)
                q_st<NQ>(Sg + ((long long)(Y0 + cl / T) * width1 + (X0 + cl % T)) * vec + s * NQ, acc[i]);
        }
        __syncthreads();
    }
    // winner-take-all over the finished S tile, one pixel per wave at a time: lane l holds dwords l*NP .. l*NP+NP-1 of
    // the vector in natural order, which live at [chunk (g % NQ) / 4][lane g / NQ][g % 4] in the tile
    int off[NP];
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        const int g = lane * NP + j, q = g % NQ;
        off[j] = (q / 4) * 64 + (g / NQ) * 4 + (q % 4);
    }
    constexpr int B = 8;                                                  // cells per batch
    for (int c0 = w * B; c0 < T * T; c0 += NW * B) {
        us2 Sv[B][NP];
#pragma unroll
        for (int u = 0; u < B; ++u)
#pragma unroll
            for (int j = 0; j < NP; ++j) Sv[u][j] = as_us2(St[(c0 + u) * vec + off[j]]);
        int res_d;
        uint32_t res_k;
        wta_batch_eval<NP, B>(Sv, lane, D, minD, uniq, res_d, res_k);
        if (lane < B) {
            const int cl = c0 + lane;
            const long long px = (long long)(Y0 + cl / T) * width1 + (X0 + cl % T);
            sel_d16[px] = (int16_t)res_d;
            sel_key[px] = res_k;
        }
    }
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
bool tile_schedule_enabled()
{
    // opt-in (WASS_AGG=tile): measured on MI355X the schedule moves 20 B/cell instead of 28 but executes 15 path steps per
    // pixel instead of 11, and these kernels are bound by VALU issue (~4.5 cycles per wave instruction), not by HBM:
    // 9.5 ms against 8.1 ms for the pair schedule at config B
    const char* agg = getenv("WASS_AGG");
    return agg && !strcmp(agg, "tile");
}

EdgeLayout edge_layout(const SgmDims& d)
{
    EdgeLayout L;
    L.T = tile_size(d.NP);
    L.ntx = (d.width1 + L.T - 1) / L.T;
    L.nty = (d.h + L.T - 1) / L.T;
    const size_t vb = (size_t)64 * d.NP * sizeof(uint32_t);
    const size_t rowb = (size_t)row_edge_vecs(L.T, d.width1, d.h) * vb, colb = (size_t)col_edge_vecs(L.T, d.width1, d.h) * vb;
    size_t off = 0;
    for (int f = 0; f < 4; ++f)
        for (int dir = 0; dir < 2; ++dir) {
            L.has[f][dir] = dir == 0 || d.ndirs == 8 || f == FAM_ROWS;
            L.off_row[f][dir] = L.off_col[f][dir] = (size_t)-1;
            if (!L.has[f][dir]) continue;
            if (f != FAM_ROWS) { L.off_row[f][dir] = off; off += (rowb + 255) & ~(size_t)255; }
            if (f != FAM_COLS) { L.off_col[f][dir] = off; off += (colb + 255) & ~(size_t)255; }
        }
    L.total = off;
    return L;
}

template <int NP>
static int launch_aggregate_tile_np(wass_ctx* c, const SgmDims& d, int* n_launches)
{
    constexpr int T = tile_size(NP);
    constexpr int U = ckpt_k(NP);
    const EdgeLayout lay = edge_layout(d);
    int rc = ensure(c, c->edges, lay.total);
    if (rc) return rc;
    if (c->debug && (rc = ensure(c, c->S, d.cells() * sizeof(uint16_t)))) return rc;
    const uint32_t* C = (const uint32_t*)c->C.p;
    EdgePtrs ep;
    for (int f = 0; f < 4; ++f)
        for (int dir = 0; dir < 2; ++dir) {
            ep.row[f][dir] = lay.off_row[f][dir] == (size_t)-1 ? nullptr : (uint32_t*)((char*)c->edges.p + lay.off_row[f][dir]);
            ep.col[f][dir] = lay.off_col[f][dir] == (size_t)-1 ? nullptr : (uint32_t*)((char*)c->edges.p + lay.off_col[f][dir]);
        }
    int nl = 0;
    // the sweeps only read C and write disjoint edge arrays: three streams keep the GPU supplied with waves from several
    // of them at once (the runtime multiplexes streams onto four hardware queues, and the context's tail / copy streams
    // need theirs: more streams here only serialise the frame pipeline)
    WASS_HIP(c, hipEventRecord(c->ev_cost, c->stream));
    constexpr int NS = 3;
    hipStream_t ss[NS] = { c->stream, c->side, c->side2 };
    hipEvent_t es[NS] = { nullptr, c->ev_ckpt[0], c->ev_ckpt[1] };
    for (int i = 1; i < NS; ++i) WASS_HIP(c, hipStreamWaitEvent(ss[i], c->ev_cost, 0));
    // longest first: the diagonal sweeps have the most chains and the tails of short ones
    static const int order[8][2] = { { FAM_DIAG, 0 }, { FAM_ANTI, 0 }, { FAM_DIAG, 1 }, { FAM_ANTI, 1 }, { FAM_ROWS, 0 },
                                     { FAM_ROWS, 1 }, { FAM_COLS, 1 }, { FAM_COLS, 0 } };
    int k = 0;
    for (const auto& o : order) {
        const int f = o[0], dir = o[1];
        if (!lay.has[f][dir]) continue;
        if (f == FAM_COLS && dir == 0) continue;              // swept by k_vsum_col while it produces C
        int dx, dy;
        family_dir(f, dx, dy);
        if (dir) { dx = -dx; dy = -dy; }
        const int nch = dy == 0 ? d.h : (dx == 0 ? d.width1 : d.width1 + d.h - 1);
        hipStream_t st = ss[(k + 1) % NS];
        if constexpr (NP <= 4) {
            constexpr int UQ = NP == 1 ? 8 : (NP == 2 ? WASS_UQ2 : 2);
            const int nwaves = quad_group_waves(dy != 0 ? d.width1 : 0, T) + quad_group_waves(dy == 0 ? d.h : (dx == 0 ? 0 : d.h - 1), T);
            hipLaunchKernelGGL((k_edge_sweep_q<NP, T, UQ>), dim3((nwaves + 3) / 4), dim3(256), 0, st, C, ep.row[f][dir],
                               ep.col[f][dir], d.width1, d.h, dx, dy, d.P1, d.P2);
        } else {
            hipLaunchKernelGGL((k_edge_sweep<NP, T, U>), dim3((nch + 3) / 4), dim3(256), 0, st, C, ep.row[f][dir], ep.col[f][dir],
                               d.width1, d.h, dx, dy, d.P1, d.P2, nch);
        }
        ++k; ++nl;
    }
    for (int i = 1; i < NS; ++i) {
        WASS_HIP(c, hipEventRecord(es[i], ss[i]));
        WASS_HIP(c, hipStreamWaitEvent(c->stream, es[i], 0));
    }
    const size_t lds = (size_t)2 * T * T * 64 * NP * sizeof(uint32_t);
    const int keep = c->debug ? 1 : 0;
    uint32_t* Sg = (uint32_t*)c->S.p;
    int16_t* sd = (int16_t*)c->sel_d16.p;
    uint32_t* sk = (uint32_t*)c->sel_key.p;
    // complete tiles: quad kernel (NP <= 4); everything else (the right / bottom border, or all tiles at NP > 4): k_tile
    int fx = -1, fy = -1;
    if constexpr (NP <= 4 && T % 4 == 0) {
        static_assert((T * T) % (8 * (T / 4)) == 0, "selection pass: whole batches per wave");
        fx = d.width1 / T; fy = d.h / T;
        if (fx > 0 && fy > 0) {
            const dim3 gq((unsigned)(fx * fy)), bq(T / 4 * 64);
            if (d.ndirs == 8) {
                WASS_HIP(c, hipFuncSetAttribute((const void*)k_tile_q<NP, T, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                hipLaunchKernelGGL((k_tile_q<NP, T, true>), gq, bq, lds, c->stream, C, ep, d.width1, d.h, fx, d.P1, d.P2, d.D, d.minD, d.uniq,
                                   keep, Sg, sd, sk);
            } else {
                WASS_HIP(c, hipFuncSetAttribute((const void*)k_tile_q<NP, T, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                hipLaunchKernelGGL((k_tile_q<NP, T, false>), gq, bq, lds, c->stream, C, ep, d.width1, d.h, fx, d.P1, d.P2, d.D, d.minD, d.uniq,
                                   keep, Sg, sd, sk);
            }
            ++nl;
        } else { fx = fy = -1; }
    }
    const int ntiles = fx < 0 ? lay.ntx * lay.nty : lay.ntx * lay.nty - fx * fy;
    if (ntiles > 0) {
        const dim3 grid((unsigned)ntiles), block(T * 64);
        if (d.ndirs == 8) {
            WASS_HIP(c, hipFuncSetAttribute((const void*)k_tile<NP, T, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL((k_tile<NP, T, true>), grid, block, lds, c->stream, C, ep, d.width1, d.h, lay.ntx, d.P1, d.P2, d.D, d.minD,
                               d.uniq, keep, Sg, sd, sk, fx, fy);
        } else {
            WASS_HIP(c, hipFuncSetAttribute((const void*)k_tile<NP, T, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL((k_tile<NP, T, false>), grid, block, lds, c->stream, C, ep, d.width1, d.h, lay.ntx, d.P1, d.P2, d.D, d.minD,
                               d.uniq, keep, Sg, sd, sk, fx, fy);
        }
    }
    ++nl;
    if (n_launches) *n_launches = nl;
    WASS_HIP(c, hipGetLastError());
    return WASS_OK;
}

int launch_aggregate_tile(wass_ctx* c, const SgmDims& d, int* n_launches)
{
    switch (d.NP) {
        case 1: return launch_aggregate_tile_np<1>(c, d, n_launches);
        case 2: return launch_aggregate_tile_np<2>(c, d, n_launches);
        case 3: return launch_aggregate_tile_np<3>(c, d, n_launches);
        case 4: return launch_aggregate_tile_np<4>(c, d, n_launches);
        case 5: return launch_aggregate_tile_np<5>(c, d, n_launches);
        case 6: return launch_aggregate_tile_np<6>(c, d, n_launches);
        case 7: return launch_aggregate_tile_np<7>(c, d, n_launches);
        case 8: return launch_aggregate_tile_np<8>(c, d, n_launches);
    }
    return set_err(c, WASS_ERR_UNSUPPORTED, "MAX_DISPARITY %d not supported (max 1024)", d.D);
}

}  // namespace wass
