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

namespace wass {

template <int NP>
__device__ __forceinline__ void sgm_step(us2 (&N)[NP], const us2 (&c)[NP], us2 (&L)[NP], const us2 P1v,
                                         const us2 P2v)
{
    // pair holding d-1 of this lane's first value / d+1 of its last value (0xFFFF outside [0,Dp))
    const uint32_t prev_last = dpp_mov<DPP_WAVE_SHR1>(0xFFFFFFFFu, as_u32(N[NP - 1]));
    const uint32_t next_first = dpp_mov<DPP_WAVE_SHL1>(0xFFFFFFFFu, as_u32(N[0]));
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        const uint32_t lo = j == 0 ? prev_last : as_u32(N[j - 1]);
        const uint32_t hi = j == NP - 1 ? next_first : as_u32(N[j + 1]);
        const us2 nl = as_us2(__builtin_amdgcn_alignbit(as_u32(N[j]), lo, 16));   // (d-1, d)
        const us2 nr = as_us2(__builtin_amdgcn_alignbit(hi, as_u32(N[j]), 16));   // (d+1, d+2)
        const us2 t = pk_min(pk_min(N[j], pk_adds(pk_min(nl, nr), P1v)), P2v);
        L[j] = pk_adds(c[j], t);
    }
    us2 m = L[0];
#pragma unroll
    for (int j = 1; j < NP; ++j) m = pk_min(m, L[j]);
    const uint32_t mn = wave_min_u32(min((uint32_t)m.x, (uint32_t)m.y));
    const us2 mv = pk_splat(mn);
#pragma unroll
    for (int j = 0; j < NP; ++j) N[j] = L[j] - mv;
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

// One path, every chain: S (+)= L_r.  FIRST: S is written, not accumulated.
template <int NP, bool FIRST, int U>
__global__ void __launch_bounds__(256) k_sweep(const uint32_t* __restrict__ C, uint32_t* __restrict__ S,
                                               int width1, int h, int dx, int dy, int P1, int P2, int nchains)
{
    const int lane = threadIdx.x & 63;
    const int c = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform
    if (c >= nchains) return;
    int x0, y0, n;
    chain_geometry(c, dx, dy, width1, h, x0, y0, n);
    const long long vec = 64 * NP;                               // dwords per pixel vector
    const long long step = ((long long)dy * width1 + dx) * vec;
    const uint32_t* cp = C + ((long long)y0 * width1 + x0) * vec + lane * NP;
    uint32_t* sp = S + ((long long)y0 * width1 + x0) * vec + lane * NP;
    const us2 P1v = pk_splat(P1), P2v = pk_splat(P2), cap = pk_splat(0x7FFF);

    us2 N[NP];
#pragma unroll
    for (int j = 0; j < NP; ++j) N[j] = pk_splat(0);

    for (int k0 = 0; k0 < n; k0 += U) {
        us2 cb[U][NP], sb[U][NP];
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (k0 + u < n) {
#pragma unroll
                for (int j = 0; j < NP; ++j) {
                    cb[u][j] = as_us2(cp[u * step + j]);
                    if (!FIRST) sb[u][j] = as_us2(sp[u * step + j]);
                }
            }
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (k0 + u < n) {
                us2 L[NP];
                sgm_step<NP>(N, cb[u], L, P1v, P2v);
#pragma unroll
                for (int j = 0; j < NP; ++j) {
                    const us2 s = FIRST ? pk_min(L[j], cap) : pk_min(pk_adds(sb[u][j], L[j]), cap);
                    sp[u * step + j] = as_u32(s);
                }
            }
        cp += U * step;
        sp += U * step;
    }
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
    const us2 P1v = pk_splat(P1), P2v = pk_splat(P2), cap = pk_splat(0x7FFF);
    const int nseg = (n + K - 1) / K;

    // ---- phase 1: checkpoints of the forward path (the last segment is never needed)
    {
        us2 N[NP];
#pragma unroll
        for (int j = 0; j < NP; ++j) N[j] = pk_splat(0);
        const uint32_t* cp = cp0;
        for (int s = 0; s < nseg - 1; ++s) {
            us2 cb[K][NP];
#pragma unroll
            for (int u = 0; u < K; ++u)
#pragma unroll
                for (int j = 0; j < NP; ++j) cb[u][j] = as_us2(cp[u * step + j]);
#pragma unroll
            for (int u = 0; u < K; ++u) {
                us2 L[NP];
                sgm_step<NP>(N, cb[u], L, P1v, P2v);
            }
#pragma unroll
            for (int j = 0; j < NP; ++j) ck[(long long)s * vec + j] = as_u32(N[j]);
            cp += K * step;
        }
    }
    // ---- phase 2: backward over the segments
    us2 Nb[NP];
#pragma unroll
    for (int j = 0; j < NP; ++j) Nb[j] = pk_splat(0);
    for (int s = nseg - 1; s >= 0; --s) {
        const int k0 = s * K;
        const int len = min(K, n - k0);
        const uint32_t* cp = cp0 + (long long)k0 * step;
        uint32_t* sp = sp0 + (long long)k0 * step;
        us2 cb[K][NP], lf[K][NP], sb[K][NP];
#pragma unroll
        for (int u = 0; u < K; ++u)
            if (u < len) {
#pragma unroll
                for (int j = 0; j < NP; ++j) {
                    cb[u][j] = as_us2(cp[u * step + j]);
                    if (SMODE != 0) sb[u][j] = as_us2(sp[u * step + j]);
                }
            }
        us2 Nf[NP];
#pragma unroll
        for (int j = 0; j < NP; ++j) Nf[j] = s == 0 ? pk_splat(0) : as_us2(ck[(long long)(s - 1) * vec + j]);
#pragma unroll
        for (int u = 0; u < K; ++u)
            if (u < len) sgm_step<NP>(Nf, cb[u], lf[u], P1v, P2v);
#pragma unroll
        for (int u = K - 1; u >= 0; --u)
            if (u < len) {
                us2 L[NP], sv[NP];
                sgm_step<NP>(Nb, cb[u], L, P1v, P2v);
#pragma unroll
                for (int j = 0; j < NP; ++j) {
                    const us2 both = pk_adds(lf[u][j], L[j]);
                    sv[j] = SMODE == 0 ? pk_min(both, cap) : pk_min(pk_adds(sb[u][j], both), cap);
                }
                if (SMODE != 2 || keepS) {
#pragma unroll
                    for (int j = 0; j < NP; ++j) sp[u * step + j] = as_u32(sv[j]);
                }
                if (SMODE == 2) {
                    const long long pix = pix0 + (long long)(k0 + u) * pixstep;
                    wta_select<NP>(sv, lane, D, minD, uniq, sel_d16 + pix, sel_key + pix);
                }
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
    constexpr int K = NP <= 2 ? 16 : (NP == 3 ? 10 : (NP == 4 ? 8 : (NP == 5 ? 6 : (NP == 6 ? 5 : 4))));
    const uint32_t* C = (const uint32_t*)c->C.p;
    uint32_t* S = (uint32_t*)c->S.p;
    int16_t* sd = (int16_t*)c->sel_d16.p;
    uint32_t* sk = (uint32_t*)c->sel_key.p;
    int nl = 0;
    auto nchains = [&](int dx, int dy) { return dy == 0 ? d.h : (dx == 0 ? d.width1 : d.width1 + d.h - 1); };
    auto maxlen = [&](int dx, int dy) { return dy == 0 ? d.width1 : (dx == 0 ? d.h : (d.width1 < d.h ? d.width1 : d.h)); };
    auto ckpt_bytes = [&](int dx, int dy) {
        return (size_t)nchains(dx, dy) * ((maxlen(dx, dy) + K - 1) / K) * (64 * NP) * sizeof(uint32_t);
    };
    size_t need = ckpt_bytes(1, 0);
    if (d.ndirs == 8) {
        need = need > ckpt_bytes(0, 1) ? need : ckpt_bytes(0, 1);
        need = need > ckpt_bytes(1, 1) ? need : ckpt_bytes(1, 1);
    }
    int rc = ensure(c, c->ckpt, need);
    if (rc) return rc;
    uint32_t* ck = (uint32_t*)c->ckpt.p;

#define WASS_PAIR(SMODE, dx, dy)                                                                             \
    do {                                                                                                     \
        const int nch = nchains(dx, dy), mseg = (maxlen(dx, dy) + K - 1) / K;                                \
        hipLaunchKernelGGL((k_pair<NP, K, SMODE>), dim3((nch + 3) / 4), dim3(256), 0, c->stream, C, S, ck,   \
                           d.width1, d.h, dx, dy, d.P1, d.P2, nch, mseg, d.D, d.minD, d.uniq,                \
                           c->debug ? 1 : 0, sd, sk);                                                        \
        ++nl;                                                                                                \
    } while (0)
#define WASS_SWEEP(FIRST, dx, dy)                                                                            \
    do {                                                                                                     \
        const int nch = nchains(dx, dy);                                                                     \
        hipLaunchKernelGGL((k_sweep<NP, FIRST, U>), dim3((nch + 3) / 4), dim3(256), 0, c->stream, C, S,      \
                           d.width1, d.h, dx, dy, d.P1, d.P2, nch);                                          \
        ++nl;                                                                                                \
    } while (0)

    if (d.ndirs == 8) {
        WASS_PAIR(0, 0, 1);      // columns:        paths 2 + 6
        WASS_PAIR(1, 1, 1);      // diagonals:      paths 1 + 7
        WASS_PAIR(1, -1, 1);     // anti-diagonals: paths 3 + 5
        WASS_PAIR(2, 1, 0);      // rows:           paths 0 + 4, winner-take-all fused
    } else {
        WASS_SWEEP(true, 0, 1);  // path 2
        WASS_SWEEP(false, 1, 1); // path 1
        WASS_SWEEP(false, -1, 1);// path 3
        WASS_PAIR(2, 1, 0);      // rows: paths 0 + 4, winner-take-all fused
    }
#undef WASS_PAIR
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
