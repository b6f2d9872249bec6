// sgm_aggregate.hip -- K3: scan-line path aggregation (the roofline kernel).
//
// Replaces the L_r recurrences inside OpenCV's computeDisparitySGBM
// (SURVEY.md Appendix A.4), reached from wass_stereo/wass_stereo.cpp:837:
//
//   L_r(p,d) = C(p,d) + min(L_r(p-r,d), L_r(p-r,d-1)+P1, L_r(p-r,d+1)+P1,
//                           min_k L_r(p-r,k)+P2) - (min_k L_r(p-r,k)+P2)
//   S(p,d)   = sat16(sum_r L_r(p,d))
//
// C is stored WITHOUT the +P2 bias, and the carried state is normalised,
// N(d) = L_r(p-r,d) - min_k L_r(p-r,k), so one step is
//   L(d) = C(d) + min(N(d), min(N(d-1), N(d+1)) + P1, P2);   N' = L - min_d L
// which is algebraically identical to the formula above whenever the int16
// precondition (A.7) holds; all values fit u16 and run on v_pk_*_u16.
//
// Parallel decomposition: every path r splits the image into independent
// chains (rows, columns, diagonals, anti-diagonals).  One wavefront owns one
// chain: the 64 lanes hold the disparity vector (2*NP values per lane), the
// d+-1 neighbours come from wave-shift DPP moves, min_d from a DPP reduction.
// Each step touches one contiguous 256*NP-byte vector of C and of S.
#include "common.h"

#include <stdlib.h>

namespace wass {

// State carried along one chain: the (un-normalised) path costs of the previous
// pixel and their minimum over d.  Keeping the minimum as a separate wave-uniform
// scalar takes its cross-lane reduction off the critical path of the next step:
//   L'(d) = C(d) + min(L(d), min(L(d-1), L(d+1)) + P1, m + P2) - m,   m' = min_d L'
// (same value as the normalised form in the header comment; only L - m matters).
template <int NP>
struct PathState {
    us2 L[NP];
    uint32_t m;
    // destinations of the two wave-shift DPP moves.  Lane 0 (resp. 63) has no source lane and keeps
    // its value, so initialising them once with 0xFFFFFFFF provides the d=-1 / d=Dp sentinels
    // without re-materialising the constant every step.
    uint32_t shr = 0xFFFFFFFFu, shl = 0xFFFFFFFFu;
    __device__ __forceinline__ void reset()
    {
#pragma unroll
        for (int j = 0; j < NP; ++j) L[j] = pk_splat(0);
        m = 0;
    }
    // checkpoint form: costs relative to their minimum
    __device__ __forceinline__ void store_normalised(uint32_t* __restrict__ p) const
    {
        const us2 mv = pk_splat(m);
#pragma unroll
        for (int j = 0; j < NP; ++j) p[j] = as_u32(L[j] - mv);
    }
    __device__ __forceinline__ void load_normalised(const us2 (&v)[NP])
    {
#pragma unroll
        for (int j = 0; j < NP; ++j) L[j] = v[j];
        m = 0;
    }
};

template <int NP>
__device__ __forceinline__ void sgm_step(PathState<NP>& st, const us2 (&c)[NP], us2 (&Lo)[NP], const us2 P1v,
                                         const uint32_t P2)
{
    // pair holding d-1 of this lane's first value / d+1 of its last value (0xFFFF outside [0,Dp))
    st.shr = dpp_mov<DPP_WAVE_SHR1>(st.shr, as_u32(st.L[NP - 1]));
    st.shl = dpp_mov<DPP_WAVE_SHL1>(st.shl, as_u32(st.L[0]));
    const uint32_t prev_last = st.shr, next_first = st.shl;
    const us2 mv = pk_splat(st.m), mp2 = pk_splat(st.m + P2);
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        const uint32_t lo = j == 0 ? prev_last : as_u32(st.L[j - 1]);
        const uint32_t hi = j == NP - 1 ? next_first : as_u32(st.L[j + 1]);
        const us2 nl = as_us2(__builtin_amdgcn_alignbit(as_u32(st.L[j]), lo, 16));   // (d-1, d)
        const us2 nr = as_us2(__builtin_amdgcn_alignbit(hi, as_u32(st.L[j]), 16));   // (d+1, d+2)
        const us2 x = pk_min(st.L[j], pk_adds(pk_min(nl, nr), P1v));
        Lo[j] = pk_adds(c[j], pk_min(x, mp2) - mv);
    }
    us2 m = Lo[0];
#pragma unroll
    for (int j = 1; j < NP; ++j) m = pk_min(m, Lo[j]);
    st.m = wave_min_u32(min((uint32_t)m.x, (uint32_t)m.y));
#pragma unroll
    for (int j = 0; j < NP; ++j) st.L[j] = Lo[j];
}

// Two independent chains advanced together, statement by statement, so that each one's dependent
// packed-math / DPP wait states are filled by the other's instructions.
template <int NP>
__device__ __forceinline__ void sgm_step_pair(PathState<NP>& a, const us2 (&ca)[NP], us2 (&La)[NP],
                                              PathState<NP>& b, const us2 (&cb)[NP], us2 (&Lb)[NP],
                                              const us2 P1v, const uint32_t P2)
{
    a.shr = dpp_mov<DPP_WAVE_SHR1>(a.shr, as_u32(a.L[NP - 1]));
    b.shr = dpp_mov<DPP_WAVE_SHR1>(b.shr, as_u32(b.L[NP - 1]));
    a.shl = dpp_mov<DPP_WAVE_SHL1>(a.shl, as_u32(a.L[0]));
    b.shl = dpp_mov<DPP_WAVE_SHL1>(b.shl, as_u32(b.L[0]));
    const us2 amv = pk_splat(a.m), amp2 = pk_splat(a.m + P2);
    const us2 bmv = pk_splat(b.m), bmp2 = pk_splat(b.m + P2);
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        const uint32_t alo = j == 0 ? a.shr : as_u32(a.L[j - 1]);
        const uint32_t blo = j == 0 ? b.shr : as_u32(b.L[j - 1]);
        const uint32_t ahi = j == NP - 1 ? a.shl : as_u32(a.L[j + 1]);
        const uint32_t bhi = j == NP - 1 ? b.shl : as_u32(b.L[j + 1]);
        const us2 anl = as_us2(__builtin_amdgcn_alignbit(as_u32(a.L[j]), alo, 16));
        const us2 bnl = as_us2(__builtin_amdgcn_alignbit(as_u32(b.L[j]), blo, 16));
        const us2 anr = as_us2(__builtin_amdgcn_alignbit(ahi, as_u32(a.L[j]), 16));
        const us2 bnr = as_us2(__builtin_amdgcn_alignbit(bhi, as_u32(b.L[j]), 16));
        us2 ax = pk_min(anl, anr);
        us2 bx = pk_min(bnl, bnr);
        ax = pk_adds(ax, P1v);
        bx = pk_adds(bx, P1v);
        ax = pk_min(a.L[j], ax);
        bx = pk_min(b.L[j], bx);
        ax = pk_min(ax, amp2);
        bx = pk_min(bx, bmp2);
        ax = ax - amv;
        bx = bx - bmv;
        La[j] = pk_adds(ca[j], ax);
        Lb[j] = pk_adds(cb[j], bx);
    }
    us2 am = La[0], bm = Lb[0];
#pragma unroll
    for (int j = 1; j < NP; ++j) { am = pk_min(am, La[j]); bm = pk_min(bm, Lb[j]); }
    uint32_t ra = min((uint32_t)am.x, (uint32_t)am.y), rb = min((uint32_t)bm.x, (uint32_t)bm.y);
    wave_min2_u32(ra, rb);
    a.m = ra; b.m = rb;
#pragma unroll
    for (int j = 0; j < NP; ++j) { a.L[j] = La[j]; b.L[j] = Lb[j]; }
}

// chain c of direction (dx,dy): start cell and length
__device__ __forceinline__ void chain_geometry(int c, int dx, int dy, int width1, int h, int& x0, int& y0, int& n)
{
    if (dy == 0) { y0 = c; x0 = dx > 0 ? 0 : width1 - 1; n = width1; }
    else if (dx == 0) { x0 = c; y0 = dy > 0 ? 0 : h - 1; n = h; }
    else {
        if (c < width1) { x0 = c; y0 = dy > 0 ? 0 : h - 1; }
        else { const int k = c - width1 + 1; x0 = dx > 0 ? 0 : width1 - 1; y0 = dy > 0 ? k : h - 1 - k; }
        const int nx = dx > 0 ? width1 - x0 : x0 + 1;
        const int ny = dy > 0 ? h - y0 : y0 + 1;
        n = min(nx, ny);
    }
}

// K consecutive vectors of a chain -> registers; GUARD: only the first len exist
template <int NP, int K, bool GUARD>
__device__ __forceinline__ void load_seg(const uint32_t* __restrict__ p, long long step, int len, us2 (&dst)[K][NP])
{
#pragma unroll
    for (int u = 0; u < K; ++u)
        if (!GUARD || u < len) {
#pragma unroll
            for (int j = 0; j < NP; ++j) dst[u][j] = as_us2(p[u * step + j]);
        }
}

template <int NP, int K>
__device__ __forceinline__ void copy_seg(us2 (&dst)[K][NP], const us2 (&src)[K][NP])
{
#pragma unroll
    for (int u = 0; u < K; ++u)
#pragma unroll
        for (int j = 0; j < NP; ++j) dst[u][j] = src[u][j];
}

// ---------------------------------------------------------------------------
// Winner-take-all on a finished S vector held in registers (Appendix A.5 steps
// 2, 3 and 5; the right-view scatter and the L-R check need the whole row and
// stay in k_lrcheck).  key = (S << 16) | d reduced with a wave minimum gives the
// smallest S and, among equals, the smallest d ("first minimum").
// ---------------------------------------------------------------------------
template <int NP>
__device__ __forceinline__ int s_at(const us2 (&Sv)[NP], int d)
{
    const int ln = d / (2 * NP), slot = d % (2 * NP);       // wave-uniform
    uint32_t pv = 0;
#pragma unroll
    for (int j = 0; j < NP; ++j)
        if ((slot >> 1) == j) pv = (uint32_t)__builtin_amdgcn_readlane((int)as_u32(Sv[j]), ln);
    return (int)((slot & 1) ? (pv >> 16) : (pv & 0xFFFF));
}

template <int NP>
__device__ __forceinline__ void wta_select(const us2 (&Sv)[NP], int lane, int D, int minD, int uniq,
                                           int16_t* __restrict__ out_d16, uint32_t* __restrict__ out_key)
{
    const int dlane = lane * 2 * NP;
    uint32_t sv[2 * NP];
#pragma unroll
    for (int j = 0; j < NP; ++j) { sv[2 * j] = Sv[j].x; sv[2 * j + 1] = Sv[j].y; }
    uint32_t key = 0xFFFFFFFFu;
#pragma unroll
    for (int j = 0; j < 2 * NP; ++j)
        if (dlane + j < D) key = min(key, (sv[j] << 16) | (uint32_t)(dlane + j));
    key = wave_min_u32(key);
    const int minS = (int)(key >> 16);
    // "if (Sval < minS)" with minS initialised to MAX_COST never fires when every S is MAX_COST
    const int best = minS >= 32767 ? -1 : (int)(key & 0xFFFF);
    bool bad = false;
#pragma unroll
    for (int j = 0; j < 2 * NP; ++j) {
        const int d = dlane + j;
        if (d < D && (int)sv[j] * (100 - uniq) < minS * 100 && abs(best - d) > 1) bad = true;
    }
    const bool reject = __any(bad);
    int out = (minD - 1) * 16;
    uint32_t k = 0xFFFFFFFFu;
    if (!reject) {                                             // wave-uniform
        int d = best;
        k = ((uint32_t)minS << 16) | (uint32_t)(best & 0xFFFF);
        if (0 < d && d < D - 1) {
            const int a = s_at<NP>(Sv, d - 1), cc = s_at<NP>(Sv, d + 1), b = minS;
            const int denom2 = max(a + cc - 2 * b, 1);
            d = d * 16 + ((a - cc) * 16 + denom2) / (denom2 * 2);
        } else
            d *= 16;
        out = d + minD * 16;
    }
    if (lane == 0) { *out_d16 = (int16_t)out; *out_key = k; }
}

// One path, every chain.  SMODE 0: S = L_r (first path)   1: S += L_r   2: last path -- S is read, finished in
// registers and handed to wta_select (stored only if keepS).  Loads of the next U steps are in flight while the
// current U steps compute.
template <int NP, int SMODE, int U>
__global__ void __launch_bounds__(256) k_sweep(const uint32_t* __restrict__ C, uint32_t* __restrict__ S,
                                               int width1, int h, int dx, int dy, int P1, int P2, int nchains, int D,
                                               int minD, int uniq, int keepS, int16_t* __restrict__ sel_d16,
                                               uint32_t* __restrict__ sel_key)
{
    const int lane = threadIdx.x & 63;
    const int c = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform
    if (c >= nchains) return;
    int x0, y0, n;
    chain_geometry(c, dx, dy, width1, h, x0, y0, n);
    const long long vec = 64 * NP;                               // dwords per pixel vector
    const long long step = ((long long)dy * width1 + dx) * vec;
    const long long pixstep = (long long)dy * width1 + dx;
    long long pix = (long long)y0 * width1 + x0;
    const uint32_t* cp = C + pix * vec + lane * NP;
    uint32_t* sp = S + pix * vec + lane * NP;
    const us2 P1v = pk_splat(P1), cap = pk_splat(0x7FFF);

    auto finish = [&](const us2 (&L)[NP], const us2 (&sin)[NP], uint32_t* so, long long px) {
        us2 sv[NP];
#pragma unroll
        for (int j = 0; j < NP; ++j) sv[j] = SMODE == 0 ? pk_min(L[j], cap) : pk_min(pk_adds(sin[j], L[j]), cap);
        if (SMODE != 2 || keepS) {
#pragma unroll
            for (int j = 0; j < NP; ++j) so[j] = as_u32(sv[j]);
        }
        if (SMODE == 2) wta_select<NP>(sv, lane, D, minD, uniq, sel_d16 + px, sel_key + px);
    };

    PathState<NP> st;
    st.reset();
    const int F = n / U, r = n - F * U;
    us2 cb[U][NP], sb[U][NP], cn[U][NP], sn[U][NP];
    if (F > 0) {
        load_seg<NP, U, false>(cp, step, U, cb);
        if (SMODE != 0) load_seg<NP, U, false>(sp, step, U, sb);
    }
    for (int g = 0; g < F; ++g) {
        if (g + 1 < F) {
            load_seg<NP, U, false>(cp + U * step, step, U, cn);
            if (SMODE != 0) load_seg<NP, U, false>(sp + U * step, step, U, sn);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            us2 L[NP];
            sgm_step<NP>(st, cb[u], L, P1v, P2);
            finish(L, sb[u], sp + u * step, pix + u * pixstep);
        }
        copy_seg<NP, U>(cb, cn);
        if (SMODE != 0) copy_seg<NP, U>(sb, sn);
        cp += U * step;
        sp += U * step;
        pix += U * pixstep;
    }
    if (r > 0) {
        load_seg<NP, U, true>(cp, step, r, cb);
        if (SMODE != 0) load_seg<NP, U, true>(sp, step, r, sb);
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (u < r) {
                us2 L[NP];
                sgm_step<NP>(st, cb[u], L, P1v, P2);
                finish(L, sb[u], sp + u * step, pix + u * pixstep);
            }
    }
}

// ---------------------------------------------------------------------------
// Two opposite paths of one chain family in one launch, S touched once.
//   phase 1: forward path over the chain; only the normalised state N at the
//            end of every K-step segment is kept (checkpoint, 1/K of a volume).
//   phase 2: the chain in reverse, one segment at a time: load its K cost
//            vectors once, recompute the forward path from the checkpoint into
//            registers, run the backward path over the same registers, and
//            add both to S.
// SMODE 0: S = Lf+Lb (first family)   1: S += Lf+Lb   2: last family -- S is
// read, finished in registers and handed to wta_select; it is stored only if
// keepS (debug fetch).
// ---------------------------------------------------------------------------
// Phase 1 of a chain-family pair: the forward path over every chain, keeping only the (normalised)
// state at the end of each K-step segment -- 1/K of a volume.  Reads C once, touches nothing else, so
// the checkpoint sweeps of all families can run concurrently with any other kernel.
template <int NP, int K>
__global__ void __launch_bounds__(256) k_ckpt(const uint32_t* __restrict__ C, uint32_t* __restrict__ ckpt,
                                              int width1, int h, int dx, int dy, int P1, int P2, int nchains,
                                              int maxseg)
{
    const int lane = threadIdx.x & 63;
    const int c = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform
    if (c >= nchains) return;
    int x0, y0, n;
    chain_geometry(c, dx, dy, width1, h, x0, y0, n);
    const long long vec = 64 * NP;
    const long long step = ((long long)dy * width1 + dx) * vec;
    const uint32_t* cp0 = C + ((long long)y0 * width1 + x0) * vec + lane * NP;
    uint32_t* ck = ckpt + ((long long)c * maxseg) * vec + lane * NP;
    const us2 P1v = pk_splat(P1);
    const int F = n / K, r = n - F * K;            // F full segments, then a tail of r steps
    const int ncp = F - (r > 0 ? 0 : 1);           // checkpoints needed: end of segments 0 .. ncp-1
    {
        PathState<NP> st;
        st.reset();
        us2 cb[K][NP], cn[K][NP];
        const uint32_t* cp = cp0;
        if (ncp > 0) load_seg<NP, K, false>(cp, step, K, cb);
        for (int s = 0; s < ncp; ++s) {
            if (s + 1 < ncp) load_seg<NP, K, false>(cp + K * step, step, K, cn);
#pragma unroll
            for (int u = 0; u < K; ++u) {
                us2 L[NP];
                sgm_step<NP>(st, cb[u], L, P1v, P2);
            }
            st.store_normalised(ck + (long long)s * vec);
            copy_seg<NP, K>(cb, cn);
            cp += K * step;
        }
    }

}

template <int NP, int K, int SMODE>
__global__ void __launch_bounds__(256) k_pair(const uint32_t* __restrict__ C, uint32_t* __restrict__ S,
                                              uint32_t* __restrict__ ckpt, int width1, int h, int dx, int dy,
                                              int P1, int P2, int nchains, int maxseg, int D, int minD, int uniq,
                                              int keepS, int16_t* __restrict__ sel_d16,
                                              uint32_t* __restrict__ sel_key)
{
    const int lane = threadIdx.x & 63;
    const int c = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform
    if (c >= nchains) return;
    int x0, y0, n;
    chain_geometry(c, dx, dy, width1, h, x0, y0, n);
    const long long vec = 64 * NP;
    const long long step = ((long long)dy * width1 + dx) * vec;
    const long long pixstep = (long long)dy * width1 + dx;
    const long long base = ((long long)y0 * width1 + x0) * vec + lane * NP;
    const long long pix0 = (long long)y0 * width1 + x0;
    const uint32_t* cp0 = C + base;
    uint32_t* sp0 = S + base;
    uint32_t* ck = ckpt + ((long long)c * maxseg) * vec + lane * NP;
    const us2 P1v = pk_splat(P1), cap = pk_splat(0x7FFF);
    const int F = n / K, r = n - F * K;            // F full segments, then a tail of r steps

    // one finished backward step: S handling + optional winner-take-all
    auto finish = [&](const us2 (&lf)[NP], const us2 (&lb)[NP], const us2 (&sin)[NP], uint32_t* sp, long long pix) {
        us2 sv[NP];
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            const us2 both = pk_adds(lf[j], lb[j]);
            sv[j] = SMODE == 0 ? pk_min(both, cap) : pk_min(pk_adds(sin[j], both), cap);
        }
        if (SMODE != 2 || keepS) {
#pragma unroll
            for (int j = 0; j < NP; ++j) sp[j] = as_u32(sv[j]);
        }
        if (SMODE == 2) wta_select<NP>(sv, lane, D, minD, uniq, sel_d16 + pix, sel_key + pix);
    };

    // ---- phase 2: the chain in reverse ------------------------------------------------------------
    PathState<NP> bw;
    bw.reset();
    if (r > 0) {                                   // tail segment F (guarded, not pipelined)
        const uint32_t* cp = cp0 + (long long)F * K * step;
        uint32_t* sp = sp0 + (long long)F * K * step;
        us2 cb[K][NP], lf[K][NP], sb[K][NP];
        load_seg<NP, K, true>(cp, step, r, cb);
        if (SMODE != 0) load_seg<NP, K, true>(sp, step, r, sb);
        PathState<NP> fw;
        fw.reset();
        if (F > 0) {
            us2 nv[NP];
#pragma unroll
            for (int j = 0; j < NP; ++j) nv[j] = as_us2(ck[(long long)(F - 1) * vec + j]);
            fw.load_normalised(nv);
        }
#pragma unroll
        for (int u = 0; u < K; ++u)
            if (u < r) sgm_step<NP>(fw, cb[u], lf[u], P1v, P2);
#pragma unroll
        for (int u = K - 1; u >= 0; --u)
            if (u < r) {
                us2 L[NP];
                sgm_step<NP>(bw, cb[u], L, P1v, P2);
                finish(lf[u], L, sb[u], sp + u * step, pix0 + (long long)(F * K + u) * pixstep);
            }
    }
    if (F == 0) return;

    // software pipeline over the full segments s = F-1 .. 0:
    //   iteration s:  backward(s)  ||  forward recompute(s-1)  ||  loads of C(s-2), S(s-1), ckpt(s-3) in flight
    us2 cA[K][NP], cB[K][NP], cC[K][NP];           // cost vectors of segments s, s-1, s-2
    us2 lA[K][NP], lB[K][NP];                      // forward path costs of segments s, s-1
    us2 sA[K][NP], sB[K][NP];                      // S of segments s, s-1
    us2 nvB[NP], nvC[NP];                          // checkpoints entering segments s-1, s-2
    {
        const int s = F - 1;
        load_seg<NP, K, false>(cp0 + (long long)s * K * step, step, K, cA);
        if (SMODE != 0) load_seg<NP, K, false>(sp0 + (long long)s * K * step, step, K, sA);
        if (s >= 1) load_seg<NP, K, false>(cp0 + (long long)(s - 1) * K * step, step, K, cB);
        PathState<NP> fw;
        fw.reset();
        if (s >= 1) {
            us2 nv[NP];
#pragma unroll
            for (int j = 0; j < NP; ++j) nv[j] = as_us2(ck[(long long)(s - 1) * vec + j]);
            fw.load_normalised(nv);
        }
#pragma unroll
        for (int j = 0; j < NP; ++j) nvB[j] = s >= 2 ? as_us2(ck[(long long)(s - 2) * vec + j]) : pk_splat(0);
#pragma unroll
        for (int u = 0; u < K; ++u) sgm_step<NP>(fw, cA[u], lA[u], P1v, P2);
    }
    for (int s = F - 1; s >= 1; --s) {
        // prefetch for the next iterations
        if (s >= 2) load_seg<NP, K, false>(cp0 + (long long)(s - 2) * K * step, step, K, cC);
        if (SMODE != 0) load_seg<NP, K, false>(sp0 + (long long)(s - 1) * K * step, step, K, sB);
#pragma unroll
        for (int j = 0; j < NP; ++j) nvC[j] = s >= 3 ? as_us2(ck[(long long)(s - 3) * vec + j]) : pk_splat(0);
        PathState<NP> fw;
        fw.load_normalised(nvB);                   // zeros when s-1 == 0
        uint32_t* sp = sp0 + (long long)s * K * step;
        const long long pixs = pix0 + (long long)s * K * pixstep;
        // two independent dependency chains in one block: the scheduler interleaves them
#pragma unroll
        for (int u = 0; u < K; ++u) {
            const int v = K - 1 - u;
            us2 L[NP];
            sgm_step_pair<NP>(fw, cB[u], lB[u], bw, cA[v], L, P1v, P2);
            finish(lA[v], L, sA[v], sp + v * step, pixs + v * pixstep);
        }
        copy_seg<NP, K>(cA, cB);
        copy_seg<NP, K>(cB, cC);
        copy_seg<NP, K>(lA, lB);
        if (SMODE != 0) copy_seg<NP, K>(sA, sB);
#pragma unroll
        for (int j = 0; j < NP; ++j) nvB[j] = nvC[j];
    }
    {                                              // epilogue: backward over segment 0
#pragma unroll
        for (int v = K - 1; v >= 0; --v) {
            us2 L[NP];
            sgm_step<NP>(bw, cA[v], L, P1v, P2);
            finish(lA[v], L, sA[v], sp0 + v * step, pix0 + v * pixstep);
        }
    }
}

template <int NP>
static int launch_aggregate_np(wass_ctx* c, const SgmDims& d, int* n_launches)
{
    // Path numbering of Appendix A.4: pass 1 = 0:(-1,0) 1:(-1,-1) 2:(0,-1) 3:(+1,-1), then 4:(+1,0);
    // MODE_HH adds 5:(-1,+1) 6:(0,+1) 7:(+1,+1).  (dx,dy) below is the direction of travel = -r.
    // Opposite paths share their chains: rows {0,4}, columns {2,6}, diagonals {1,7}, anti-diagonals {3,5}.
    constexpr int U = NP <= 2 ? 8 : (NP <= 4 ? 4 : 2);
    constexpr int K = NP <= 2 ? 8 : (NP <= 4 ? 4 : 2);
    const uint32_t* C = (const uint32_t*)c->C.p;
    uint32_t* S = (uint32_t*)c->S.p;
    int16_t* sd = (int16_t*)c->sel_d16.p;
    uint32_t* sk = (uint32_t*)c->sel_key.p;
    int nl = 0;
    auto nchains = [&](int dx, int dy) { return dy == 0 ? d.h : (dx == 0 ? d.width1 : d.width1 + d.h - 1); };
    auto maxlen = [&](int dx, int dy) { return dy == 0 ? d.width1 : (dx == 0 ? d.h : (d.width1 < d.h ? d.width1 : d.h)); };
    auto ckpt_bytes = [&](int dx, int dy) {
        const size_t b = (size_t)nchains(dx, dy) * ((maxlen(dx, dy) + K - 1) / K) * (64 * NP) * sizeof(uint32_t);
        return (b + 255) & ~(size_t)255;
    };
    // families in launch order; every family has its own checkpoint region so that all checkpoint
    // sweeps (which only read C) can run ahead on the side stream while the main stream accumulates S
    struct Fam { int dx, dy, smode; };
    Fam fam[4];
    int nf = 0;
    // Order: the kernel that carries the winner-take-all goes last and should have the most chains (the WTA adds
    // ~50 instructions per pixel): the anti-diagonals (width1 + h - 1 chains).
    fam[nf++] = { 1, 0, 0 };         // rows:           paths 0 + 4   (S written)
    if (d.ndirs == 8) {
        fam[nf++] = { 0, 1, 1 };     // columns:        paths 2 + 6
        fam[nf++] = { 1, 1, 1 };     // diagonals:      paths 1 + 7
        fam[nf++] = { -1, 1, 2 };    // anti-diagonals: paths 3 + 5, winner-take-all fused
    }
    size_t off[5] = { 0 };
    for (int f = 0; f < nf; ++f) off[f + 1] = off[f] + ckpt_bytes(fam[f].dx, fam[f].dy);
    int rc = ensure(c, c->ckpt, off[nf]);
    if (rc) return rc;

    WASS_HIP(c, hipEventRecord(c->ev_cost, c->stream));
    WASS_HIP(c, hipStreamWaitEvent(c->side, c->ev_cost, 0));
    WASS_HIP(c, hipStreamWaitEvent(c->side2, c->ev_cost, 0));
    for (int f = 0; f < nf; ++f) {
        const int dx = fam[f].dx, dy = fam[f].dy;
        const int nch = nchains(dx, dy), mseg = (maxlen(dx, dy) + K - 1) / K;
        static const bool two = getenv("WASS_SIDE_STREAMS") && atoi(getenv("WASS_SIDE_STREAMS")) == 2;
        hipStream_t ss = (two && (f & 1)) ? c->side2 : c->side;
        hipLaunchKernelGGL((k_ckpt<NP, K>), dim3((nch + 3) / 4), dim3(256), 0, ss, C,
                           (uint32_t*)((char*)c->ckpt.p + off[f]), d.width1, d.h, dx, dy, d.P1, d.P2, nch, mseg);
        WASS_HIP(c, hipEventRecord(c->ev_ckpt[f], ss));
        ++nl;
    }

    for (int f = 0; f < nf; ++f) {
        const int dx = fam[f].dx, dy = fam[f].dy;
        const int nch = nchains(dx, dy), mseg = (maxlen(dx, dy) + K - 1) / K;
        uint32_t* ck = (uint32_t*)((char*)c->ckpt.p + off[f]);
        WASS_HIP(c, hipStreamWaitEvent(c->stream, c->ev_ckpt[f], 0));
        const dim3 grid((nch + 3) / 4), block(256);
#define WASS_PAIR(SMODE)                                                                                     \
        hipLaunchKernelGGL((k_pair<NP, K, SMODE>), grid, block, 0, c->stream, C, S, ck, d.width1, d.h, dx, dy, \
                           d.P1, d.P2, nch, mseg, d.D, d.minD, d.uniq, c->debug ? 1 : 0, sd, sk)
        if (fam[f].smode == 0) WASS_PAIR(0);
        else if (fam[f].smode == 1) WASS_PAIR(1);
        else WASS_PAIR(2);
#undef WASS_PAIR
        ++nl;
    }
#define WASS_SWEEP(SMODE, dx, dy)                                                                            \
    do {                                                                                                     \
        const int nch = nchains(dx, dy);                                                                     \
        hipLaunchKernelGGL((k_sweep<NP, SMODE, U>), dim3((nch + 3) / 4), dim3(256), 0, c->stream, C, S,      \
                           d.width1, d.h, dx, dy, d.P1, d.P2, nch, d.D, d.minD, d.uniq, c->debug ? 1 : 0, sd, sk); \
        ++nl;                                                                                                \
    } while (0)
    if (d.ndirs == 5) {              // MODE_SGBM: the three down-going paths have no partner
        WASS_SWEEP(1, 0, 1);         // path 2
        WASS_SWEEP(1, 1, 1);         // path 1
        WASS_SWEEP(2, -1, 1);        // path 3, winner-take-all fused
    }
#undef WASS_SWEEP
    if (n_launches) *n_launches = nl;
    WASS_HIP(c, hipGetLastError());
    return WASS_OK;
}

int launch_aggregate(wass_ctx* c, const SgmDims& d, int* n_launches)
{
    switch (d.NP) {
        case 1: return launch_aggregate_np<1>(c, d, n_launches);
        case 2: return launch_aggregate_np<2>(c, d, n_launches);
        case 3: return launch_aggregate_np<3>(c, d, n_launches);
        case 4: return launch_aggregate_np<4>(c, d, n_launches);
        case 5: return launch_aggregate_np<5>(c, d, n_launches);
        case 6: return launch_aggregate_np<6>(c, d, n_launches);
        case 7: return launch_aggregate_np<7>(c, d, n_launches);
        case 8: return launch_aggregate_np<8>(c, d, n_launches);
    }
    return set_err(c, WASS_ERR_UNSUPPORTED, "MAX_DISPARITY %d not supported (max 1024)", d.D);
}

}  // namespace wass
