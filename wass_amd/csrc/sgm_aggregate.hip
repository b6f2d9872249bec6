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
#include "sgm_step.h"

#include <stdlib.h>

namespace wass {

// One path, every chain.  SMODE 0: S = L_r (first path)   1: S += L_r   2: last path -- S is read, finished in
// registers and handed to wta_batch (stored only if keepS).  Loads of the next U steps are in flight while the
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

    // finished S of one step; in the last sweep the U vectors of a group are kept for wta_batch
    auto finish = [&](const us2 (&L)[NP], const us2 (&sin)[NP], uint32_t* so, us2 (&sv)[NP]) {
#pragma unroll
        for (int j = 0; j < NP; ++j) sv[j] = SMODE == 0 ? pk_min(L[j], cap) : pk_min(pk_adds(sin[j], L[j]), cap);
        if (SMODE != 2 || keepS) {
            st_stream_vec<NP>(so, sv);
        }
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
        us2 fin[U][NP];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            us2 L[NP];
            sgm_step<NP>(st, cb[u], L, P1v, P2);
            finish(L, sb[u], sp + u * step, fin[u]);
        }
        if (SMODE == 2) wta_batch<NP, U>(fin, U, lane, D, minD, uniq, sel_d16, sel_key, pix, pixstep);
        copy_seg<NP, U>(cb, cn);
        if (SMODE != 0) copy_seg<NP, U>(sb, sn);
        cp += U * step;
        sp += U * step;
        pix += U * pixstep;
    }
    if (r > 0) {
        load_seg<NP, U, true>(cp, step, r, cb);
        if (SMODE != 0) load_seg<NP, U, true>(sp, step, r, sb);
        us2 fin[U][NP];
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
            for (int j = 0; j < NP; ++j) fin[u][j] = cap;
            if (u < r) {
                us2 L[NP];
                sgm_step<NP>(st, cb[u], L, P1v, P2);
                finish(L, sb[u], sp + u * step, fin[u]);
            }
        }
        if (SMODE == 2) wta_batch<NP, U>(fin, r, lane, D, minD, uniq, sel_d16, sel_key, pix, pixstep);
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
// read, finished in registers and handed to wta_batch; it is stored only if
// keepS (debug fetch).
// ---------------------------------------------------------------------------
// Phase 1 of a chain-family pair: the forward path over every chain, keeping only the (normalised)
// state at the end of each K-step segment -- 1/K of a volume.  Reads C once, touches nothing else, so
// the checkpoint sweeps of all families can run concurrently with any other kernel.
// endstate != nullptr: the family is split in the middle (half_chain_geometry): c counts sub-chains, the sweep runs to the
// end of its half and leaves its final state in endstate[c] for the pair kernel of the other half.
template <int NP, int K>
__global__ void __launch_bounds__(256) k_ckpt(const uint32_t* __restrict__ C, uint32_t* __restrict__ ckpt,
                                              int width1, int h, int dx, int dy, int P1, int P2, int nchains,
                                              int maxseg, uint32_t* __restrict__ endstate)
{
    const int lane = threadIdx.x & 63;
    const int c = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform
    if (c >= nchains) return;
    int x0, y0, n;
    if (endstate) half_chain_geometry(c, dx, dy, width1, h, x0, y0, n);
    else chain_geometry(c, dx, dy, width1, h, x0, y0, n);
    const long long vec = 64 * NP;
    const long long step = ((long long)dy * width1 + dx) * vec;
    const uint32_t* cp0 = C + ((long long)y0 * width1 + x0) * vec + lane * NP;
    uint32_t* ck = ckpt + ((long long)c * maxseg) * vec + lane * NP;
    const us2 P1v = pk_splat(P1);
    const int F = n / K, r = n - F * K;            // F full segments, then a tail of r steps
    // checkpoints needed: end of segments 0 .. ncp-1 (a split family also needs the state at the very end)
    const int ncp = endstate ? F : F - (r > 0 ? 0 : 1);
    {
        PathState<NP> st;
        st.reset();
        us2 cb[K][NP], cn[K][NP];
        const uint32_t* cp = cp0;
        if (ncp > 0) load_seg<NP, K, false>(cp, step, K, cb);
        for (int s = 0; s < ncp; ++s) {
            if (s + 1 < ncp) load_seg<NP, K, false>(cp + K * step, step, K, cn);
            else if (endstate && r > 0) load_seg<NP, K, true>(cp + K * step, step, r, cn);
#pragma unroll
            for (int u = 0; u < K; ++u) {
                us2 L[NP];
                sgm_step<NP>(st, cb[u], L, P1v, P2);
            }
            st.store_normalised(ck + (long long)s * vec);
            copy_seg<NP, K>(cb, cn);
            cp += K * step;
        }
        if (endstate) {
            if (ncp == 0 && r > 0) load_seg<NP, K, true>(cp, step, r, cb);
#pragma unroll
            for (int u = 0; u < K; ++u)
                if (u < r) {
                    us2 L[NP];
                    sgm_step<NP>(st, cb[u], L, P1v, P2);
                }
            st.store_normalised(endstate + (long long)c * vec + lane * NP);
        }
    }

}

// SMODE 3: like 2, but the partial sums of the earlier sweeps arrive in two volumes (S + S2).
// SMODE 4: like 1 with two inputs: S = S + S2 + Lf + Lb.
template <int NP, int K, int SMODE>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4))) k_pair(const uint32_t* __restrict__ C, uint32_t* __restrict__ S,
                                              const uint32_t* __restrict__ S2,
                                              uint32_t* __restrict__ ckpt, int width1, int h, int dx, int dy,
                                              int P1, int P2, int nchains, int maxseg, int D, int minD, int uniq,
                                              int keepS, int16_t* __restrict__ sel_d16,
                                              uint32_t* __restrict__ sel_key, const uint32_t* __restrict__ endstate)
{
    const int lane = threadIdx.x & 63;
    const int c = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform
    if (c >= nchains) return;
    int x0, y0, n;
    if (endstate) half_chain_geometry(c, dx, dy, width1, h, x0, y0, n);    // split family: c counts sub-chains (k_ckpt)
    else chain_geometry(c, dx, dy, width1, h, x0, y0, n);
    const long long vec = 64 * NP;
    const long long step = ((long long)dy * width1 + dx) * vec;
    const long long pixstep = (long long)dy * width1 + dx;
    const long long base = ((long long)y0 * width1 + x0) * vec + lane * NP;
    const long long pix0 = (long long)y0 * width1 + x0;
    const uint32_t* cp0 = C + base;
    uint32_t* sp0 = S + base;
    const uint32_t* tp0 = S2 + base;
    uint32_t* ck = ckpt + ((long long)c * maxseg) * vec + lane * NP;
    const us2 P1v = pk_splat(P1), cap = pk_splat(0x7FFF);
    const int F = n / K, r = n - F * K;            // F full segments, then a tail of r steps
    constexpr bool LAST = SMODE == 2 || SMODE == 3, TWO = SMODE == 3 || SMODE == 4;

    // one finished backward step: S handling + optional winner-take-all
    auto finish = [&](const us2 (&lf)[NP], const us2 (&lb)[NP], const us2 (&sin)[NP], const us2 (&sin2)[NP], uint32_t* sp,
                      us2 (&sv)[NP]) {
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            const us2 both = pk_adds(lf[j], lb[j]);
            us2 acc = SMODE == 0 ? both : pk_adds(sin[j], both);
            if (TWO) acc = pk_adds(acc, sin2[j]);
            sv[j] = pk_min(acc, cap);
        }
        if (!LAST || keepS) {
            st_stream_vec<NP>(sp, sv);
        }
    };

    // ---- phase 2: the chain in reverse ------------------------------------------------------------
    PathState<NP> bw;
    bw.reset();
    if (endstate) {                                // the backward path arrives from the other half of the chain
        us2 nv[NP];
        ld_stream_vec<NP>(endstate + (long long)(c ^ 1) * vec + lane * NP, nv);
        bw.load_normalised(nv);
    }
    if (r > 0) {                                   // tail segment F (guarded, not pipelined)
        const uint32_t* cp = cp0 + (long long)F * K * step;
        uint32_t* sp = sp0 + (long long)F * K * step;
        us2 cb[K][NP], lf[K][NP], sb[K][NP], tb[K][NP];
        load_seg<NP, K, true>(cp, step, r, cb);
        if (SMODE != 0) load_seg<NP, K, true>(sp, step, r, sb);
        if (TWO) load_seg<NP, K, true>(tp0 + (long long)F * K * step, step, r, tb);
        PathState<NP> fw;
        fw.reset();
        if (F > 0) {
            us2 nv[NP];
            ld_stream_vec<NP>(ck + (long long)(F - 1) * vec, nv);
            fw.load_normalised(nv);
        }
#pragma unroll
        for (int u = 0; u < K; ++u)
            if (u < r) sgm_step<NP>(fw, cb[u], lf[u], P1v, P2);
        us2 fin[K][NP];
#pragma unroll
        for (int u = K - 1; u >= 0; --u) {
#pragma unroll
            for (int j = 0; j < NP; ++j) fin[u][j] = cap;
            if (u < r) {
                us2 L[NP];
                sgm_step<NP>(bw, cb[u], L, P1v, P2);
                finish(lf[u], L, sb[u], tb[u], sp + u * step, fin[u]);
            }
        }
        if (LAST) wta_batch<NP, K>(fin, r, lane, D, minD, uniq, sel_d16, sel_key, pix0 + (long long)F * K * pixstep, pixstep);
    }
    if (F == 0) return;

    // software pipeline over the full segments s = F-1 .. 0:
    //   iteration s:  backward(s)  ||  forward recompute(s-1)  ||  loads of C(s-2), S(s-1), ckpt(s-3) in flight
    us2 cA[K][NP], cB[K][NP], cC[K][NP];           // cost vectors of segments s, s-1, s-2
    us2 lA[K][NP], lB[K][NP];                      // forward path costs of segments s, s-1
    us2 sA[K][NP], sB[K][NP];                      // S of segments s, s-1
    us2 tA[K][NP], tB[K][NP];                      // S2 of segments s, s-1 (SMODE 3)
    us2 nvB[NP], nvC[NP];                          // checkpoints entering segments s-1, s-2
    {
        const int s = F - 1;
        load_seg<NP, K, false>(cp0 + (long long)s * K * step, step, K, cA);
        if (SMODE != 0) load_seg<NP, K, false>(sp0 + (long long)s * K * step, step, K, sA);
        if (TWO) load_seg<NP, K, false>(tp0 + (long long)s * K * step, step, K, tA);
        if (s >= 1) load_seg<NP, K, false>(cp0 + (long long)(s - 1) * K * step, step, K, cB);
        PathState<NP> fw;
        fw.reset();
        if (s >= 1) {
            us2 nv[NP];
            ld_stream_vec<NP>(ck + (long long)(s - 1) * vec, nv);
            fw.load_normalised(nv);
        }
#pragma unroll
        for (int j = 0; j < NP; ++j) nvB[j] = s >= 2 ? as_us2(ld_stream(ck + (long long)(s - 2) * vec + j)) : pk_splat(0);
#pragma unroll
        for (int u = 0; u < K; ++u) sgm_step<NP>(fw, cA[u], lA[u], P1v, P2);
    }
    for (int s = F - 1; s >= 1; --s) {
        // prefetch for the next iterations
        if (s >= 2) load_seg<NP, K, false>(cp0 + (long long)(s - 2) * K * step, step, K, cC);
        if (SMODE != 0) load_seg<NP, K, false>(sp0 + (long long)(s - 1) * K * step, step, K, sB);
        if (TWO) load_seg<NP, K, false>(tp0 + (long long)(s - 1) * K * step, step, K, tB);
#pragma unroll
        for (int j = 0; j < NP; ++j) nvC[j] = s >= 3 ? as_us2(ld_stream(ck + (long long)(s - 3) * vec + j)) : pk_splat(0);
        PathState<NP> fw;
        fw.load_normalised(nvB);                   // zeros when s-1 == 0
        uint32_t* sp = sp0 + (long long)s * K * step;
        const long long pixs = pix0 + (long long)s * K * pixstep;
        // two independent dependency chains in one block: the scheduler interleaves them
        us2 fin[K][NP];
#pragma unroll
        for (int u = 0; u < K; ++u) {
            const int v = K - 1 - u;
            us2 L[NP];
            sgm_step_pair<NP>(fw, cB[u], lB[u], bw, cA[v], L, P1v, P2);
            finish(lA[v], L, sA[v], tA[v], sp + v * step, fin[v]);
        }
        if (LAST) wta_batch<NP, K>(fin, K, lane, D, minD, uniq, sel_d16, sel_key, pixs, pixstep);
        copy_seg<NP, K>(cA, cB);
        copy_seg<NP, K>(cB, cC);
        copy_seg<NP, K>(lA, lB);
        if (SMODE != 0) copy_seg<NP, K>(sA, sB);
        if (TWO) copy_seg<NP, K>(tA, tB);
#pragma unroll
        for (int j = 0; j < NP; ++j) nvB[j] = nvC[j];
    }
    {                                              // epilogue: backward over segment 0
        us2 fin[K][NP];
#pragma unroll
        for (int v = K - 1; v >= 0; --v) {
            us2 L[NP];
            sgm_step<NP>(bw, cA[v], L, P1v, P2);
            finish(lA[v], L, sA[v], tA[v], sp0 + v * step, fin[v]);
        }
        if (LAST) wta_batch<NP, K>(fin, K, lane, D, minD, uniq, sel_d16, sel_key, pix0, pixstep);
    }
}

CkptLayout ckpt_layout(const SgmDims& d)
{
    CkptLayout L;
    L.K = ckpt_k(d.NP);
    const char* agg = getenv("WASS_AGG");
    const bool legacy = agg && (!strcmp(agg, "trio") || !strcmp(agg, "concurrent") || !strcmp(agg, "rowsfirst"));
    const char* sp = getenv("WASS_SPLIT_ROWS");
    const bool split_rows = !legacy && (!sp || atoi(sp) != 0);
    const char* sdg = getenv("WASS_SPLIT_DIAG");
    const bool split_diag = !legacy && (!sdg || atoi(sdg) != 0);
    const char* sc = getenv("WASS_SPLIT_COLS");
    const bool split_cols = !legacy && d.ndirs == 8 && (!sc || atoi(sc) != 0);      // only the family that k_vsum_col produces
    auto add = [&](int dx, int dy, int smode) {
        const int f = L.nfam++;
        L.dx[f] = dx; L.dy[f] = dy; L.smode[f] = smode;
        L.nch[f] = dy == 0 ? d.h : (dx == 0 ? d.width1 : d.width1 + d.h - 1);
        int maxlen = dy == 0 ? d.width1 : (dx == 0 ? d.h : (d.width1 < d.h ? d.width1 : d.h));
        // rows: 2 058 chains are two waves per SIMD -- split them in the middle (half_chain_geometry) for twice the waves.
        // Measured at config B, same box, three runs each: columns split 1.40 -> 1.11 ms (k_vsum_col), diagonals split
        // 7.90 -> 7.60 ms (aggregation): twice the waves, and the longest chain of a diagonal family halves.
        L.split[f] = (split_rows && dy == 0) || (split_cols && dx == 0) || (split_diag && dx != 0 && dy != 0);
        if (L.split[f]) { L.nch[f] *= 2; maxlen = maxlen - maxlen / 2; }
        L.mseg[f] = (maxlen + L.K - 1) / L.K;
        const size_t b = (size_t)L.nch[f] * (L.mseg[f] + (L.split[f] ? 1 : 0)) * (64 * d.NP) * sizeof(uint32_t);   // + the end states
        L.off[f + 1] = L.off[f] + ((b + 255) & ~(size_t)255);
    };
    // The kernel that carries the winner-take-all goes last and should have the most chains (the WTA adds ~60
    // instructions per pixel): the anti-diagonals (width1 + h - 1 chains).
    if (d.ndirs == 8 && !legacy) {
        L.cols_from_cost = true;
        add(0, 1, 0);                // columns:        paths 2 + 6   (S written)
        add(1, 0, 1);                // rows:           paths 0 + 4
        add(1, 1, 1);                // diagonals:      paths 1 + 7
        add(-1, 1, 2);               // anti-diagonals: paths 3 + 5, winner-take-all fused
    } else if (d.ndirs == 5 && !legacy) {
        L.path2_from_cost = true;
        add(1, 0, 1);                // rows: paths 0 + 4, added to the S = L_2 that the cost stage left behind
    } else {
        add(1, 0, 0);                // rows (S written)
        if (d.ndirs == 8) { add(0, 1, 1); add(1, 1, 1); add(-1, 1, 2); }
    }
    return L;
}

// Pipelined-strip schedule (sgm_trio.hip): paths {0,1,2} and {4,7,6} (MODE_HH) / {4,3} (MODE_SGBM) as two
// concurrent three-path sweeps writing S and S2; MODE_HH finishes with the anti-diagonal pair {3,5} + fused
// winner-take-all reading both volumes, MODE_SGBM with a plain S + S2 selection kernel.
template <int NP>
static int launch_aggregate_trio(wass_ctx* c, const SgmDims& d, int* n_launches)
{
    constexpr int K = ckpt_k(NP);
    const uint32_t* C = (const uint32_t*)c->C.p;
    uint32_t* S = (uint32_t*)c->S.p;
    int rc;
    const size_t vol = d.cells() * sizeof(uint16_t), hb = trio_halo_bytes(d);
    if ((rc = ensure(c, c->S2, vol))) return rc;
    // boundary buffers of the two sweeps; (re)allocation or a previous time-out leaves them in an unknown state
    const size_t before = c->halo.cap;
    if ((rc = ensure(c, c->halo, 2 * hb))) return rc;
    if (c->halo.cap != before || c->halo_dirty) {
        WASS_HIP(c, hipMemsetAsync(c->halo.p, 0xFF, c->halo.cap, c->stream));
        c->halo_dirty = false;
    }
    uint32_t* S2 = (uint32_t*)c->S2.p;
    unsigned long long* haloA = (unsigned long long*)c->halo.p;
    unsigned long long* haloB = (unsigned long long*)((char*)c->halo.p + hb);
    int nl = 0;
    WASS_HIP(c, hipEventRecord(c->ev_cost, c->stream));
    WASS_HIP(c, hipStreamWaitEvent(c->side, c->ev_cost, 0));
    if ((rc = launch_trio(c, d, S, haloA, +1, +1, true, c->stream))) return rc;                   // paths 0, 1, 2 -> S
    ++nl;
    if (d.ndirs == 8) {
        if ((rc = launch_trio(c, d, S2, haloB, -1, -1, true, c->side))) return rc;               // paths 4, 7, 6 -> S2
        ++nl;
        WASS_HIP(c, hipEventRecord(c->ev_ckpt[0], c->side));
        // anti-diagonals: checkpoint sweep on the second side stream, then the pair + selection
        WASS_HIP(c, hipStreamWaitEvent(c->side2, c->ev_cost, 0));
        const int nch = d.width1 + d.h - 1, mlen = d.width1 < d.h ? d.width1 : d.h, mseg = (mlen + K - 1) / K;
        const size_t cb = (size_t)nch * mseg * (64 * NP) * sizeof(uint32_t);
        if ((rc = ensure(c, c->ckpt, cb))) return rc;
        uint32_t* ck = (uint32_t*)c->ckpt.p;
        hipLaunchKernelGGL((k_ckpt<NP, K>), dim3((nch + 3) / 4), dim3(256), 0, c->side2, C, ck, d.width1, d.h, -1, 1, d.P1, d.P2,
                           nch, mseg, (uint32_t*)nullptr);
        ++nl;
        WASS_HIP(c, hipEventRecord(c->ev_ckpt[1], c->side2));
        WASS_HIP(c, hipStreamWaitEvent(c->stream, c->ev_ckpt[0], 0));
        WASS_HIP(c, hipStreamWaitEvent(c->stream, c->ev_ckpt[1], 0));
        hipLaunchKernelGGL((k_pair<NP, K, 3>), dim3((nch + 3) / 4), dim3(256), 0, c->stream, C, S, (const uint32_t*)S2, ck,
                           d.width1, d.h, -1, 1, d.P1, d.P2, nch, mseg, d.D, d.minD, d.uniq, c->debug ? 1 : 0,
                           (int16_t*)c->sel_d16.p, (uint32_t*)c->sel_key.p, (const uint32_t*)nullptr);
        ++nl;
    } else {
        if ((rc = launch_trio(c, d, S2, haloB, -1, +1, false, c->side))) return rc;              // paths 4, 3 -> S2
        ++nl;
        WASS_HIP(c, hipEventRecord(c->ev_ckpt[0], c->side));
        WASS_HIP(c, hipStreamWaitEvent(c->stream, c->ev_ckpt[0], 0));
        if ((rc = launch_wta_sum(c, d, S, S2, c->stream))) return rc;
        ++nl;
    }
    if (n_launches) *n_launches = nl;
    WASS_HIP(c, hipGetLastError());
    return WASS_OK;
}

template <int NP>
static int launch_aggregate_np(wass_ctx* c, const SgmDims& d, int* n_launches)
{
    // Path numbering of Appendix A.4: pass 1 = 0:(-1,0) 1:(-1,-1) 2:(0,-1) 3:(+1,-1), then 4:(+1,0);
    // MODE_HH adds 5:(-1,+1) 6:(0,+1) 7:(+1,+1).  (dx,dy) below is the direction of travel = -r.
    // Opposite paths share their chains: rows {0,4}, columns {2,6}, diagonals {1,7}, anti-diagonals {3,5}.
    constexpr int U = ckpt_k(NP);
    constexpr int K = ckpt_k(NP);
    const uint32_t* C = (const uint32_t*)c->C.p;
    uint32_t* S = (uint32_t*)c->S.p;
    int16_t* sd = (int16_t*)c->sel_d16.p;
    uint32_t* sk = (uint32_t*)c->sel_key.p;
    int nl = 0;
    // WASS_AGG=trio selects the pipelined column-strip schedule (sgm_trio.hip).  It moves 40 % fewer bytes and is
    // bit-exact, but measured slower on MI355X (10-12 ms vs 8.1 ms at config B): a strip is ONE wave walking
    // 2058 rows x 12 dependent path steps, and a lone wave issues only ~1 VALU instruction per 16 cycles.
    const char* agg = getenv("WASS_AGG");
    if (agg && !strcmp(agg, "trio")) return launch_aggregate_trio<NP>(c, d, n_launches);
    auto nchains = [&](int dx, int dy) { return dy == 0 ? d.h : (dx == 0 ? d.width1 : d.width1 + d.h - 1); };
    // every family has its own checkpoint region so that all checkpoint sweeps (which only read C) can run ahead
    // on the side stream while the main stream accumulates S
    const CkptLayout lay = ckpt_layout(d);
    const int nf = lay.nfam;
    struct Fam { int dx, dy, smode; };
    Fam fam[4];
    size_t off[5];
    for (int f = 0; f < nf; ++f) { fam[f] = { lay.dx[f], lay.dy[f], lay.smode[f] }; off[f] = lay.off[f]; }
    off[nf] = lay.off[nf];
    int rc = ensure(c, c->ckpt, off[nf]);
    if (rc) return rc;

    WASS_HIP(c, hipEventRecord(c->ev_cost, c->stream));
    WASS_HIP(c, hipStreamWaitEvent(c->side, c->ev_cost, 0));
    WASS_HIP(c, hipStreamWaitEvent(c->side2, c->ev_cost, 0));
    for (int f = 0; f < nf; ++f) {
        if (f == 0 && lay.cols_from_cost) {         // written by k_vsum_col on the main stream already
            WASS_HIP(c, hipEventRecord(c->ev_ckpt[f], c->stream));
            continue;
        }
        const int dx = fam[f].dx, dy = fam[f].dy;
        const int nch = lay.nch[f], mseg = lay.mseg[f];
        static const bool two = getenv("WASS_SIDE_STREAMS") && atoi(getenv("WASS_SIDE_STREAMS")) == 2;
        hipStream_t ss = (two && (f & 1)) ? c->side2 : c->side;
        uint32_t* ckf = (uint32_t*)((char*)c->ckpt.p + off[f]);
        hipLaunchKernelGGL((k_ckpt<NP, K>), dim3((nch + 3) / 4), dim3(256), 0, ss, C, ckf, d.width1, d.h, dx, dy, d.P1, d.P2, nch, mseg,
                           lay.split[f] ? ckf + (size_t)nch * mseg * (64 * NP) : (uint32_t*)nullptr);
        WASS_HIP(c, hipEventRecord(c->ev_ckpt[f], ss));
        ++nl;
    }

    // WASS_AGG=concurrent (experiment): the row and the column family, which have the fewest chains, run
    // concurrently, each writing its own volume (S, S2), and the first diagonal family folds the two together.
    // Same bytes as the serial order and, measured, the same time (8.3 vs 8.2 ms): the family is bandwidth-bound.
    const bool conc = d.ndirs == 8 && getenv("WASS_AGG") && !strcmp(getenv("WASS_AGG"), "concurrent");
    if (conc) {
        int rc2 = ensure(c, c->S2, d.cells() * sizeof(uint16_t));
        if (rc2) return rc2;
    }
    const uint32_t* S2 = conc ? (const uint32_t*)c->S2.p : (const uint32_t*)S;
    if (lay.path2_from_cost) {       // path 1 needs no checkpoints: it runs while the row checkpoints are produced
        const int nch = nchains(1, 1);
        hipLaunchKernelGGL((k_sweep<NP, 1, U>), dim3((nch + 3) / 4), dim3(256), 0, c->stream, C, S, d.width1, d.h, 1, 1, d.P1, d.P2,
                           nch, d.D, d.minD, d.uniq, c->debug ? 1 : 0, sd, sk);
        ++nl;
    }
    for (int f = 0; f < nf; ++f) {
        const int dx = fam[f].dx, dy = fam[f].dy;
        const int nch = lay.nch[f], mseg = lay.mseg[f];
        uint32_t* ck = (uint32_t*)((char*)c->ckpt.p + off[f]);
        const dim3 grid((nch + 3) / 4), block(256);
#define WASS_PAIR(SMODE, STREAM, SOUT)                                                                       \
        hipLaunchKernelGGL((k_pair<NP, K, SMODE>), grid, block, 0, STREAM, C, SOUT, S2, ck, d.width1, d.h, dx, dy, \
                           d.P1, d.P2, nch, mseg, d.D, d.minD, d.uniq, c->debug ? 1 : 0, sd, sk,                \
                           lay.split[f] ? (const uint32_t*)(ck + (size_t)nch * mseg * (64 * NP)) : (const uint32_t*)nullptr)
        if (conc && f == 1) {                       // columns -> S2 on the second side stream, concurrent with the rows
            WASS_HIP(c, hipStreamWaitEvent(c->side2, c->ev_ckpt[f], 0));
            WASS_PAIR(0, c->side2, (uint32_t*)c->S2.p);
            WASS_HIP(c, hipEventRecord(c->ev_cols, c->side2));
        } else {
            WASS_HIP(c, hipStreamWaitEvent(c->stream, c->ev_ckpt[f], 0));
            if (conc && f == 2) {                   // diagonals: S = S + S2 + pair
                WASS_HIP(c, hipStreamWaitEvent(c->stream, c->ev_cols, 0));
                WASS_PAIR(4, c->stream, S);
            } else if (fam[f].smode == 0) WASS_PAIR(0, c->stream, S);
            else if (fam[f].smode == 1) WASS_PAIR(1, c->stream, S);
            else WASS_PAIR(2, c->stream, S);
        }
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
    if (d.ndirs == 5 && !lay.path2_from_cost) {   // MODE_SGBM, legacy order: the three down-going paths as plain sweeps
        WASS_SWEEP(1, 0, 1);         // path 2
        WASS_SWEEP(1, 1, 1);         // path 1
        WASS_SWEEP(2, -1, 1);        // path 3, winner-take-all fused
    } else if (d.ndirs == 5) {
        WASS_SWEEP(2, -1, 1);        // path 3, winner-take-all fused (paths 2, 1, 0 + 4 are in S by now)
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
