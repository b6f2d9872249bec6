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

template <int NP>
static int launch_aggregate_np(wass_ctx* c, const SgmDims& d, int* n_launches)
{
    // path order of Appendix A.4: pass 1 = 0:(-1,0) 1:(-1,-1) 2:(0,-1) 3:(+1,-1), then 4:(+1,0);
    // MODE_HH adds 5:(-1,+1) 6:(0,+1) 7:(+1,+1).  (dx,dy) below is the direction of travel = -r.
    static const int DIRS[8][2] = { { 1, 0 }, { 1, 1 }, { 0, 1 }, { -1, 1 }, { -1, 0 }, { 1, -1 }, { 0, -1 }, { -1, -1 } };
    constexpr int U = NP <= 2 ? 8 : (NP <= 4 ? 4 : 2);
    for (int r = 0; r < d.ndirs; ++r) {
        const int dx = DIRS[r][0], dy = DIRS[r][1];
        const int nch = dy == 0 ? d.h : (dx == 0 ? d.width1 : d.width1 + d.h - 1);
        dim3 grid((nch + 3) / 4);
        if (r == 0)
            hipLaunchKernelGGL((k_sweep<NP, true, U>), grid, dim3(256), 0, c->stream, (const uint32_t*)c->C.p,
                               (uint32_t*)c->S.p, d.width1, d.h, dx, dy, d.P1, d.P2, nch);
        else
            hipLaunchKernelGGL((k_sweep<NP, false, U>), grid, dim3(256), 0, c->stream, (const uint32_t*)c->C.p,
                               (uint32_t*)c->S.p, d.width1, d.h, dx, dy, d.P1, d.P2, nch);
    }
    if (n_launches) *n_launches = d.ndirs;
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
