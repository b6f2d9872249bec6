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

namespace wass {

// One path, every chain (5-path mode: paths 1 and 3, which have no partner).  SMODE 0: S = L_r (first path)
// 1: S += L_r   2: last path -- S is read, finished in registers and handed to wta_batch (stored only if keepS).
// Loads of the next U steps are in flight while the current U steps compute.
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
// Two opposite paths of one chain family, S touched once (k_ckpt + k_pair).
//   k_ckpt: forward path over the chain; only the normalised state N at the
//           end of every K-step segment is kept (checkpoint, 1/K of a volume),
//           plus the scalar min_d L after every step (2 bytes per pixel).
//   k_pair: the chain in reverse, one segment at a time: recompute the forward
//           path from the checkpoint into registers, run the backward path over
//           the same cost vectors, and add both to S.
// ---------------------------------------------------------------------------
// Reads C once, touches nothing else, so the checkpoint sweeps of all families run ahead on the side stream while the
// main stream accumulates S.  The K cost vectors of a segment sit in a register ring that is refilled in place: the
// slot of step u is reloaded with step u of the NEXT segment as soon as it has been consumed (prefetch distance exactly
// K steps, no second buffer, no register copies).
// endstate != nullptr: the family is split in the middle (half_chain_geometry): c counts sub-chains, the sweep runs to the
// end of its half and leaves its final state in endstate[c] for the pair kernel of the other half.
// At most WASS_CKPT_WAVES waves per SIMD: a sweep needs 26 VGPRs and would otherwise take every wave slot the pair kernel
// leaves (4 + 4 of 8), which locks the previous frame's tail kernels out of the GPU for as long as both run.
#ifndef WASS_CKPT_WAVES
#define WASS_CKPT_WAVES 3
#endif
template <int NP, int K>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, WASS_CKPT_WAVES))) k_ckpt(const uint32_t* __restrict__ C, uint32_t* __restrict__ ckpt,
                                              uint16_t* __restrict__ mins, int width1, int h, int dx, int dy, int P1, int P2,
                                              int nchains, int maxseg, uint32_t* __restrict__ endstate)
{
    const int lane = threadIdx.x & 63;
    const int c = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform
    if (c >= nchains) return;
    int x0, y0, n;
    if (endstate) half_chain_geometry(c, dx, dy, width1, h, x0, y0, n);
    else chain_geometry(c, dx, dy, width1, h, x0, y0, n);
    constexpr int VB = 256 * NP;                   // bytes per pixel vector
    const long long vec = 64 * NP;
    ChainAddr a;
    a.pixstep = (long long)dy * width1 + dx;
    a.pix0 = (long long)y0 * width1 + x0;
    a.sstep = (int)a.pixstep * VB;
    const uint32_t voff = lane * NP * 4;
    const uint32_t bK = a.bias(K);
    const us2 P1v = pk_splat(P1);
    const int F = n / K, r = n - F * K;            // F full segments, then a tail of r steps
    // checkpoints needed: end of segments 0 .. ncp-1 (a split family also needs the state at the very end)
    const int ncp = endstate ? F : F - (r > 0 ? 0 : 1);
    const rsrc_t ckr = mk_rsrc(ckpt + (long long)c * maxseg * vec);
    uint32_t* mrow = (uint32_t*)(mins + (size_t)c * maxseg * K);

    PathState<NP> st;
    st.reset();
    us2 ring[K][NP];
    if (ncp > 0) {
        const rsrc_t r0 = a.run<NP>(C, 0, K);
#pragma unroll
        for (int u = 0; u < K; ++u) buf_ld<NP>(r0, voff, bK + u * a.sstep, ring[u]);
    }
    for (int s = 0; s < ncp; ++s) {
        uint32_t ms[K];                            // min_d L after every step of the segment
        // refill from the next full segment; past the last one the slots are re-read from it (never consumed)
        const rsrc_t rn = a.run<NP>(C, (long long)min(s + 1, F - 1) * K, K);
#pragma unroll
        for (int u = 0; u < K; ++u) {
            us2 L[NP];
            sgm_step<NP>(st, ring[u], L, P1v, P2);
            ms[u] = st.m;
            buf_ld<NP>(rn, voff, bK + u * a.sstep, ring[u]);
        }
        {
            const us2 mvv = pk_splat(st.m);
            us2 nrm[NP];
#pragma unroll
            for (int j = 0; j < NP; ++j) nrm[j] = st.L[j] - mvv;
            buf_st<NP>(ckr, voff, (uint32_t)s * VB, nrm);
        }
        store_minima<K>(mrow + (size_t)s * (K / 2), ms, lane);
    }
    if (endstate) {
        if (r > 0) {
            const rsrc_t rt = a.run<NP>(C, (long long)F * K, r);
            const uint32_t br = a.bias(r);
#pragma unroll
            for (int u = 0; u < K; ++u)
                if (u < r) buf_ld<NP>(rt, voff, br + u * a.sstep, ring[u]);
#pragma unroll
            for (int u = 0; u < K; ++u)
                if (u < r) {
                    us2 L[NP];
                    sgm_step<NP>(st, ring[u], L, P1v, P2);
                }
        }
        st.store_normalised(endstate + (long long)c * vec + lane * NP);
    }
}

// ---------------------------------------------------------------------------
// k_pair: on-chip storage of one wave
//   registers: two rings that are refilled IN PLACE, exactly K steps ahead (no staging buffer, no register copies,
//     one loop shape): cf[u] -- the cost vector the forward recomputation consumes at step u; it is reloaded with
//     element u of the next (lower) segment as soon as it has been used; sr[v] -- S of the element the backward path is
//     at, reloaded with the same element of the next segment.
//   LDS (8 KiB per wave, private to the wave: no barriers): the hand-over from the forward recomputation to the backward
//     path, which visits the same segment one iteration later in the opposite order: K cost vectors and K forward path
//     cost vectors.  The slot the backward path has just drained at step u (element K-1-u of segment s) receives
//     element u of segment s-1, so the slot order flips every iteration -- in LDS that is an address, not a register
//     assignment: two iteration shapes (ODD = slot order reversed) instead of a rotating register file.
// ---------------------------------------------------------------------------
template <int NP>
__device__ __forceinline__ void lds_st(uint32_t* p, const us2 (&v)[NP])
{
    if constexpr (NP % 4 == 0) {
#pragma unroll
        for (int j = 0; j < NP; j += 4) *(wass_u32x4*)(p + j) = wass_u32x4{ as_u32(v[j]), as_u32(v[j + 1]), as_u32(v[j + 2]), as_u32(v[j + 3]) };
    } else if constexpr (NP % 2 == 0) {
#pragma unroll
        for (int j = 0; j < NP; j += 2) *(wass_u32x2*)(p + j) = wass_u32x2{ as_u32(v[j]), as_u32(v[j + 1]) };
    } else {
#pragma unroll
        for (int j = 0; j < NP; ++j) p[j] = as_u32(v[j]);
    }
}
template <int NP>
__device__ __forceinline__ void lds_ld(const uint32_t* p, us2 (&v)[NP])
{
    if constexpr (NP % 4 == 0) {
#pragma unroll
        for (int j = 0; j < NP; j += 4) {
            const wass_u32x4 t = *(const wass_u32x4*)(p + j);
            v[j] = as_us2(t.x); v[j + 1] = as_us2(t.y); v[j + 2] = as_us2(t.z); v[j + 3] = as_us2(t.w);
        }
    } else if constexpr (NP % 2 == 0) {
#pragma unroll
        for (int j = 0; j < NP; j += 2) {
            const wass_u32x2 t = *(const wass_u32x2*)(p + j);
            v[j] = as_us2(t.x); v[j + 1] = as_us2(t.y);
        }
    } else {
#pragma unroll
        for (int j = 0; j < NP; ++j) v[j] = as_us2(p[j]);
    }
}

// A minima record (N u16 values, N / 2 dwords) is held in ONE VGPR, dword i in lane i, and read back with v_readlane.
// Scalar loads would be the obvious form, but SMEM returns out of order: while one is outstanding every wait for an LDS
// read becomes lgkmcnt(0), so the first hand-over read of an iteration stalled for a full HBM round trip (k_pairx, cycle
// counters: column phase 12.3k cycles per iteration in the waves that had just requested their row records, 6.0k in the
// two waves that had not -- and the other eight waiting for them at the barrier).  Vector loads count on vmcnt, in order,
// behind ring refills that are waited for anyway.
// A minima record (N u16 values, N / 2 dwords).  VEC = false: N / 2 SGPRs filled by scalar loads.  VEC = true: ONE VGPR,
// dword i in lane i, read back with v_readlane.  SMEM returns out of order, so while a scalar load is outstanding every
// wait for an LDS read becomes lgkmcnt(0); in k_pairx, where a wave requests its row records right before the column
// phase, the first hand-over read of that phase then stalls for a full HBM round trip -- and nine other waves wait for it
// at the barrier.  Vector loads count on vmcnt, in order, behind ring refills that are waited for anyway.
// The compiler fence in load() keeps the vector load where it is written.  Without it k_pair<2, 8, 1> was MISCOMPILED
// (rocm 7.2 clang, -O3: the load is sunk past the loop-exit test into the next half-iteration and S comes out wrong in
// segment 0 of short diagonal chains -- 232 cells of the 320 x 64, D = 256 test; bit-exact again with the fence, with any
// of -mllvm -disable-machine-sink, -mllvm -disable-machine-licm, -O1, or with a scalar load of the same record kept alive
// beside the vector one; forcing every s_waitcnt to zero did NOT help, so it is not a missing wait, and scripts/micro/
// order.hip shows vmcnt counting in order across global / buffer / scratch loads and stores.  DESIGN.md 4.3).
#ifndef WASS_VREC
#define WASS_VREC 7                                // bit 0: k_pair, bit 1: k_pairx column records, bit 2: k_pairx row records
#endif
#ifndef WASS_REC_TIE
#define WASS_REC_TIE 0
#endif
#ifndef WASS_REC_FENCE
#define WASS_REC_FENCE 1                           // 0: the build that miscompiles k_pair<2, 8, 1> -- kept selectable so that the
#endif                                             // device self-test (wass_sgm_selftest) can be shown to catch it
template <int N, bool VEC>
struct Rec {
    uint32_t v[VEC ? 1 : N / 2];
    __device__ __forceinline__ void clear()
    {
#pragma unroll
        for (int i = 0; i < (VEC ? 1 : N / 2); ++i) v[i] = 0;
    }
    __device__ __forceinline__ void load(const uint32_t* p, int lane)
    {
        if constexpr (VEC) {
            // wave-uniform base in a descriptor + a small per-lane offset: no 64-bit per-lane address to keep alive
            v[0] = __builtin_amdgcn_raw_buffer_load_b32(mk_rsrc(p), (uint32_t)min(lane, N / 2 - 1) * 4u, 0, 0);
            // the loaded REGISTER is an operand of the fence: a bare memory clobber does not formally order a read-only
            // buffer-load intrinsic, a volatile asm that consumes its result does (it cannot be sunk past it)
#if WASS_REC_TIE
            asm volatile("" : "+v"(v[0]) :: "memory");
#elif WASS_REC_FENCE
            asm volatile("" ::: "memory");
#endif
        } else {
#pragma unroll
            for (int i = 0; i < N / 2; ++i) v[i] = p[i];
        }
    }
    __device__ __forceinline__ uint32_t at(int i) const
    {
        if constexpr (VEC) return (__builtin_amdgcn_readlane(v[0], i >> 1) >> ((i & 1) * 16)) & 0xFFFFu;
        else return (v[i >> 1] >> ((i & 1) * 16)) & 0xFFFFu;
    }
    __device__ __forceinline__ uint32_t at_dyn(int i) const        // i: wave-uniform, not a compile-time constant
    {
        uint32_t w;
        if constexpr (VEC) w = __builtin_amdgcn_readlane(v[0], i >> 1);
        else {
            w = v[0];
#pragma unroll
            for (int k = 1; k < N / 2; ++k) w = (i >> 1) == k ? v[k] : w;
        }
        return (w >> ((i & 1) * 16)) & 0xFFFFu;
    }
    // minimum of the forward path costs BEFORE step u of a segment: 0 for the normalised checkpoint state, otherwise
    // what the checkpoint sweep recorded after step u - 1
    __device__ __forceinline__ uint32_t before(int u) const { return u == 0 ? 0u : at(u - 1); }
};

// Occupancy is capped at four waves per SIMD (the hand-over buffers of four workgroups are 128 of the CU's 160 KiB of LDS):
// with a fifth workgroup per CU the frame tail of the previous frame, which runs underneath on its own stream and needs
// LDS for its tile kernels, starves until a pair kernel drains (measured: frame period 9.7 ms against an SGM stage of 8.7).
#ifndef WASS_PAIR_WAVES
#define WASS_PAIR_WAVES 4
#endif
// SMODE 0: S = Lf+Lb (first family)   1: S += Lf+Lb   2: last family -- S is read, finished (sat16) in registers and
// handed to wta_batch; it is stored only if keepS (debug fetch).
template <int NP, int K, int SMODE>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WASS_PAIR_WAVES, WASS_PAIR_WAVES)))
k_pair(const uint32_t* __restrict__ C, uint32_t* __restrict__ S, const uint32_t* __restrict__ ckpt,
       const uint16_t* __restrict__ mins, int width1, int h, int dx, int dy, int P1, int P2, int nchains, int maxseg, int D,
       int minD, int uniq, int keepS, int16_t* __restrict__ sel_d16, uint32_t* __restrict__ sel_key,
       const uint32_t* __restrict__ endstate)
{
    static_assert(K % 2 == 0, "the minima records are read as dwords");
    constexpr int VW = 64 * NP;                    // dwords per pixel vector
    __shared__ __attribute__((aligned(16))) uint32_t hand[4][2][K][VW];   // per wave: [0] cost vectors, [1] forward path costs
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = blockIdx.x * 4 + wv;             // wave-uniform
    if (c >= nchains) return;
    int x0, y0, n;
    if (endstate) half_chain_geometry(c, dx, dy, width1, h, x0, y0, n);    // split family: c counts sub-chains (k_ckpt)
    else chain_geometry(c, dx, dy, width1, h, x0, y0, n);
    constexpr int VB = 256 * NP;
    constexpr bool LAST = SMODE == 2;
    const long long vec = VW;
    ChainAddr a;
    a.pixstep = (long long)dy * width1 + dx;
    a.pix0 = (long long)y0 * width1 + x0;
    a.sstep = (int)a.pixstep * VB;
    const uint32_t voff = lane * NP * 4;
    const uint32_t bK = a.bias(K);
    const us2 P1v = pk_splat(P1), cap = pk_splat(0x7FFF);
    const int F = n / K, r = n - F * K;            // F full segments, then a tail of r steps
    const rsrc_t ckr = mk_rsrc(ckpt + (long long)c * maxseg * vec);
    const uint32_t* mrow = (const uint32_t*)(mins + (size_t)c * maxseg * K);       // K / 2 dwords per segment
    uint32_t* hc = &hand[wv][0][0][lane * NP];     // slot e: hc + e * VW
    uint32_t* hl = &hand[wv][1][0][lane * NP];

    PathState<NP> bw;
    bw.reset();
    if (endstate) {                                // the backward path arrives from the other half of the chain
        us2 nv[NP];
        ld_stream_vec<NP>(endstate + (long long)(c ^ 1) * vec + lane * NP, nv);
        bw.load_normalised(nv);
    }
    // S of one finished backward step
    auto finish = [&](const us2 (&lf)[NP], const us2 (&lb)[NP], const us2 (&sin)[NP], us2 (&sv)[NP]) {
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            const us2 both = pk_adds(lf[j], lb[j]);
            const us2 acc = SMODE == 0 ? both : pk_adds(sin[j], both);
            sv[j] = LAST ? pk_min(acc, cap) : acc;                 // sat16 once, at the end: every term is >= 0
        }
    };
    if (r > 0) {                                   // tail segment F (guarded, not pipelined)
        const rsrc_t rc = a.run<NP>(C, (long long)F * K, r), rs = a.run<NP>(S, (long long)F * K, r);
        const uint32_t br = a.bias(r);
        us2 cb[K][NP], lf[K][NP], sb[K][NP];
#pragma unroll
        for (int u = 0; u < K; ++u)
            if (u < r) {
                buf_ld<NP>(rc, voff, br + u * a.sstep, cb[u]);
                if (SMODE != 0) buf_ld<NP>(rs, voff, br + u * a.sstep, sb[u]);
            }
        PathState<NP> fw;
        fw.reset();
        if (F > 0) {
            us2 nv[NP];
            buf_ld<NP>(ckr, voff, (uint32_t)(F - 1) * VB, nv);
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
                finish(lf[u], L, sb[u], fin[u]);
                if (!LAST || keepS) buf_st<NP>(rs, voff, br + u * a.sstep, fin[u]);
            }
        }
        if (LAST) wta_batch<NP, K>(fin, r, lane, D, minD, uniq, sel_d16, sel_key, a.pix0 + (long long)F * K * a.pixstep, a.pixstep);
    }
    if (F == 0) return;

    // prologue: the forward path over segment F-1 (with its reductions: once per chain) into the hand-over slots in
    // natural order, segment F-2 into the forward ring, the first S vectors
    us2 cf[K][NP], sr[K][NP];
    PathState<NP> fw;
    us2 nvB[NP];                                   // checkpoint entering the segment the next forward recomputation covers
    Rec<K, (WASS_VREC & 1) != 0> mB;               // ... and that segment's minima record
    mB.clear();
    {
        const int s = F - 1;
        const rsrc_t rc = a.run<NP>(C, (long long)s * K, K), rs = a.run<NP>(S, (long long)s * K, K);
        const rsrc_t rc2 = a.run<NP>(C, (long long)max(s - 1, 0) * K, K);
        // what the LOOP consumes first is requested first: vmcnt counts in order, and the waits inside the loop are computed
        // for the worst of its predecessors (with these two loads last, the loop began every iteration with vmcnt(1))
        if (s >= 2) buf_ld<NP>(ckr, voff, (uint32_t)(s - 2) * VB, nvB);
        else {
#pragma unroll
            for (int j = 0; j < NP; ++j) nvB[j] = pk_splat(0);
        }
        mB.load(mrow + (size_t)max(s - 1, 0) * (K / 2), lane);
        us2 c0[K][NP];
        fw.reset();
        us2 nv[NP];
        if (s >= 1) buf_ld<NP>(ckr, voff, (uint32_t)(s - 1) * VB, nv);
#pragma unroll
        for (int u = 0; u < K; ++u) {
            buf_ld<NP>(rc, voff, bK + u * a.sstep, c0[u]);
            if (SMODE != 0) buf_ld<NP>(rs, voff, bK + u * a.sstep, sr[u]);
        }
#pragma unroll
        for (int u = 0; u < K; ++u) buf_ld<NP>(rc2, voff, bK + u * a.sstep, cf[u]);
        if (s >= 1) fw.load_normalised(nv);
#pragma unroll
        for (int u = 0; u < K; ++u) {
            us2 L[NP];
            sgm_step<NP>(fw, c0[u], L, P1v, P2);
            lds_st<NP>(hc + u * VW, c0[u]);
            lds_st<NP>(hl + u * VW, L);
        }
    }

    // One iteration for segment s >= 1: backward path over segment s (elements K-1 .. 0, read from the hand-over slots) ||
    // forward recomputation of segment s-1 (elements 0 .. K-1, from the ring; results into the slots just drained) ||
    // refills: C of segment s-2, S of segment s-1.  odd: the slots hold segment s in reversed order; the direction in which
    // they are walked is a run-time stride, so that the loop has ONE body: with two compile-time forms unrolled behind each
    // other, the refills of the first had their only use behind the loop-exit test between them and the compiler sank
    // them there -- K steps too late, the second form began by waiting for loads it had just issued.
    us2 fin[K][NP];
    bool odd = false;
    int s = F - 1;
    // Nothing in flight when the loop is entered: its waits are then the ones its own back-edge needs (with
    // the prologue's loads pending the top of every trip waited for vmcnt(4) instead of 9 at K = 4; at K = 8 it stays at
    // vmcnt(8) because the scheduler issues the refills in clumps of four or five.  Pinning them to their steps with
    // sched_barrier costs more than it gains: 6.42 against 6.36 ms).
    __builtin_amdgcn_s_waitcnt(0x0F70);                    // vmcnt(0), expcnt and lgkmcnt untouched
    while (s >= 1) {
        const rsrc_t rcN = a.run<NP>(C, (long long)max(s - 2, 0) * K, K);      // past the chain start: a harmless re-read
        const rsrc_t rsN = a.run<NP>(S, (long long)(s - 1) * K, K), rsO = a.run<NP>(S, (long long)s * K, K);
        us2 nvC[NP];
        Rec<K, (WASS_VREC & 1) != 0> mC;
        if (s >= 3) buf_ld<NP>(ckr, voff, (uint32_t)(s - 3) * VB, nvC);
        else {
#pragma unroll
            for (int j = 0; j < NP; ++j) nvC[j] = pk_splat(0);
        }
        mC.load(mrow + (size_t)max(s - 2, 0) * (K / 2), lane);
        fw.load_normalised(nvB);                   // zeros when s-1 == 0
        const int stp = odd ? VW : -VW;
        uint32_t* pc = hc + (odd ? 0 : (K - 1) * VW);
        us2 cbn[NP], lfn[NP];                      // hand-over vectors of the NEXT step, in flight
        lds_ld<NP>(pc, cbn);
        lds_ld<NP>(pc + K * VW, lfn);              // hl = hc + K * VW
#pragma unroll
        for (int u = 0; u < K; ++u) {
            const int v = K - 1 - u;               // element of segment s the backward path is at
            uint32_t* pn = pc + stp;
            us2 cb[NP], lfv[NP], Lf[NP], Lb[NP], sv[NP];
#pragma unroll
            for (int j = 0; j < NP; ++j) { cb[j] = cbn[j]; lfv[j] = lfn[j]; }
            if (u + 1 < K) {
                lds_ld<NP>(pn, cbn);
                lds_ld<NP>(pn + K * VW, lfn);
            }
            sgm_step_fb<NP>(fw, mB.before(u), cf[u], Lf, bw, cb, Lb, P1v, P2);
            finish(lfv, Lb, sr[v], sv);
            if (!LAST || keepS) buf_st<NP>(rsO, voff, bK + v * a.sstep, sv);
            if (LAST) {
#pragma unroll
                for (int j = 0; j < NP; ++j) fin[v][j] = sv[j];
            }
            lds_st<NP>(pc, cf[u]);
            lds_st<NP>(pc + K * VW, Lf);
            buf_ld<NP>(rcN, voff, bK + u * a.sstep, cf[u]);
            if (SMODE != 0) buf_ld<NP>(rsN, voff, bK + v * a.sstep, sr[v]);
            pc = pn;
        }
        if (LAST) wta_batch<NP, K>(fin, K, lane, D, minD, uniq, sel_d16, sel_key, a.pix0 + (long long)s * K * a.pixstep, a.pixstep);
#pragma unroll
        for (int j = 0; j < NP; ++j) nvB[j] = nvC[j];
        mB = mC;
        odd = !odd;
        --s;
    }
    {   // the backward path over segment 0: nothing left to recompute or to prefetch
        const rsrc_t rsO = a.run<NP>(S, 0, K);
        const int stp = odd ? VW : -VW;
        uint32_t* pc = hc + (odd ? 0 : (K - 1) * VW);
#pragma unroll
        for (int u = 0; u < K; ++u) {
            const int v = K - 1 - u;
            us2 cb[NP], lfv[NP], Lb[NP], sv[NP];
            lds_ld<NP>(pc, cb);
            lds_ld<NP>(pc + K * VW, lfv);
            sgm_step<NP>(bw, cb, Lb, P1v, P2);
            finish(lfv, Lb, sr[v], sv);
            if (!LAST || keepS) buf_st<NP>(rsO, voff, bK + v * a.sstep, sv);
            if (LAST) {
#pragma unroll
                for (int j = 0; j < NP; ++j) fin[v][j] = sv[j];
            }
            pc += stp;
        }
        if (LAST) wta_batch<NP, K>(fin, K, lane, D, minD, uniq, sel_d16, sel_key, a.pix0, a.pixstep);
    }
}

// ---------------------------------------------------------------------------
// 8-path mode: the ROW family rides on the column family's pair kernel, so that S is written once for both (an S
// read-modify-write pass less: -4 B/cell of HBM traffic, -2 B/cell of writes).
//   k_rowsweep (x2, pure read streams): paths 0 and 4 over every row, keeping only the state with which the path enters
//     each block of XB columns and min_d L after every step.
//   k_pairx: k_pair of the column family with a workgroup of XB ADJACENT columns walking in lock-step.  What a wave hands
//     from its forward recomputation to its backward path (cost vectors + forward path costs of a K-row segment, in LDS)
//     is, over the XB waves, a K x XB block of the image: between the two uses wave r takes ROW r of the block, rebuilds
//     both row paths across the XB columns from the entry states (no reductions: the minima are recorded) and adds
//     L_0 + L_4 into the forward path costs waiting in LDS.  Two workgroup barriers per iteration.
// ---------------------------------------------------------------------------
// XB = 10: 2 * ceil(2455 / 10) = 492 workgroups of ten waves, two per CU (2 x 80 KiB of LDS, five waves per SIMD): all of
// config B's columns are resident at once.  With XB = 8 there were 614 workgroups for 512 places and the kernel ran two
// rounds, the second one a fifth full (2.85 ms).
// D = 257 .. 512 (NP = 3, 4; round 4): the hand-over block of XB columns x K = 8 rows x two kinds is XB * 16 vectors of 768 B /
// 1 KiB -- 120 KiB with ten columns at NP = 3, 128 KiB with EIGHT columns at NP = 4: one workgroup per CU (2.5 / 2 waves per
// SIMD, which is also all the plain pair kernel gets there: its hand-over slots are 16 KiB per wave), every wave with a row of
// its own in the row phase.  The earlier attempt at D = 512 kept ten columns and halved K to fit (two barriers every FOUR
// rows, six of ten waves idle in the row phase: 28.1 against 23.3 ms); with K = 8 the fused form is the one S pass less that
// the byte budget of config E needed.
#ifndef WASS_FUSE_NP
#define WASS_FUSE_NP 4
#endif
template <int NP> struct Fuse {
    static constexpr int XB = NP <= 3 ? 10 : 8;              // columns per workgroup of k_pairx = block size of the row entry states
    static constexpr int MINW = NP <= 2 ? 5 : 2;              // waves per SIMD the register budget is cut for
    static constexpr int MAXW = NP <= 2 ? 5 : (NP == 3 ? 3 : 2);
};
inline int fuse_xb(int NP) { return NP <= 3 ? 10 : 8; }

// One wave per row walks BOTH paths at once, path 0 from the left border and path 4 from the right border, the two
// recurrences interleaved statement by statement (sgm_step_pair): the row chains are the longest in the image (2 455
// dependent steps) and the main stream waits for this kernel, so what counts is the latency of a step, and a second,
// independent chain fills the wait states of the first (one chain per wave: 1.2 ms; this form: see DESIGN.md).
//   entF[y][b] / entB[y][b]: normalised state with which path 0 / path 4 enters block b of XB columns (undefined for the
//   block the path starts in);  MF[y][x] / MB[y][x]: min_d L after the step at x.
template <int NP>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 4)))
k_rowsweep(const uint32_t* __restrict__ C, uint32_t* __restrict__ entF, uint32_t* __restrict__ entB, uint16_t* __restrict__ MF,
           uint16_t* __restrict__ MB, int width1, int h, int P1, int P2, int nbx)
{
    constexpr int XB = Fuse<NP>::XB;
    const int lane = threadIdx.x & 63;
    const int y = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform
    if (y >= h) return;
    constexpr int VB = 256 * NP;
    const long long vec = 64 * NP;
    const int n = width1;
    ChainAddr af, ab;                                       // step t: path 0 is at x = t, path 4 at x = n-1-t
    af.pixstep = 1;  af.pix0 = (long long)y * width1;           af.sstep = VB;
    ab.pixstep = -1; ab.pix0 = (long long)y * width1 + n - 1;   ab.sstep = -VB;
    const uint32_t voff = lane * NP * 4;
    const us2 P1v = pk_splat(P1);
    const int cb = n % XB;                                  // path 4 enters a new block at the steps t = cb (mod XB)
    uint32_t* mfrow = (uint32_t*)(MF + (size_t)y * nbx * XB);
    uint16_t* mbrow = MB + (size_t)y * nbx * XB;
    uint32_t* efrow = entF + (size_t)y * nbx * vec + lane * NP;
    uint32_t* ebrow = entB + (size_t)y * nbx * vec + lane * NP;

    PathState<NP> sf, sb;
    sf.reset();
    sb.reset();
    us2 rf[XB][NP], rb[XB][NP];
    const int G = n / XB, rem = n - G * XB;                 // G groups of XB steps, then rem steps
    if (G > 0) {
        const rsrc_t r0 = af.run<NP>(C, 0, XB), r1 = ab.run<NP>(C, 0, XB);
        const uint32_t b1 = ab.bias(XB);
#pragma unroll
        for (int u = 0; u < XB; ++u) {
            buf_ld<NP>(r0, voff, u * VB, rf[u]);
            buf_ld<NP>(r1, voff, b1 + u * ab.sstep, rb[u]);
        }
        // Nothing in flight when the loop is entered ("waits are computed for the worst predecessor"): the
        // scheduler reorders these twenty loads, the loop's first operands came out 7 and 3 from the end and every trip
        // began with s_waitcnt vmcnt(7) / vmcnt(3); now vmcnt(19) / vmcnt(18) at every step.  (No measurable change in
        // this kernel: what its two entry-state stores per block cost -- 1.41 ms against 1.08 without them, same
        // arithmetic -- is not a wait at this point; ring depth 20 / 30, stores without nt, through a descriptor, to one
        // address, or batched at the end of a block all leave it where it is.  Keeping the stored registers alive for a
        // block gives 0.1 ms of it back in isolation and nothing in the frame.)
        __builtin_amdgcn_s_waitcnt(0x0F70);                 // vmcnt(0), expcnt and lgkmcnt untouched
    }
    for (int g = 0; g < G; ++g) {
        const int t0 = g * XB;
        // refill from the next group; steps past the end of the row re-read its last pixel and are never consumed
        const int nxt = min(t0 + XB, n - 1), cnn = min(XB, n - nxt);
        const rsrc_t rnf = af.run<NP>(C, nxt, cnn), rnb = ab.run<NP>(C, nxt, cnn);
        const uint32_t bnb = ab.bias(cnn);
        uint32_t msf[XB], msb[XB];
#pragma unroll
        for (int u = 0; u < XB; ++u) {
            const int t = t0 + u;
            if (u == 0 && g > 0) sf.store_normalised(efrow + (size_t)g * vec);                  // path 0 enters block g
            if (u == cb && t > 0) sb.store_normalised(ebrow + (size_t)((n - 1 - t) / XB) * vec); // path 4 enters block (n-1-t)/XB
            us2 Lf[NP], Lb[NP];
            sgm_step_pair<NP>(sf, rf[u], Lf, sb, rb[u], Lb, P1v, P2);
            msf[u] = sf.m;
            msb[XB - 1 - u] = sb.m;                         // in x order: x = n-1-t0-u
            const int uc = min(u, cnn - 1);
            buf_ld<NP>(rnf, voff, uc * VB, rf[u]);
            buf_ld<NP>(rnb, voff, bnb + uc * ab.sstep, rb[u]);
        }
        store_minima<XB>(mfrow + (size_t)g * (XB / 2), msf, lane);
        {                                                   // path 4's minima cover x = n-t0-XB .. n-1-t0: not block aligned
            uint32_t v = msb[0];
#pragma unroll
            for (int i = 1; i < XB; ++i) v = lane == i ? msb[i] : v;
            if (lane < XB) mbrow[n - t0 - XB + lane] = (uint16_t)v;
        }
    }
    if (rem > 0) {                                          // the last rem steps, guarded
        const int t0 = G * XB;
        if (G == 0) {
            const rsrc_t r0 = af.run<NP>(C, 0, rem), r1 = ab.run<NP>(C, 0, rem);
            const uint32_t b1 = ab.bias(rem);
#pragma unroll
            for (int u = 0; u < XB; ++u)
                if (u < rem) {
                    buf_ld<NP>(r0, voff, u * VB, rf[u]);
                    buf_ld<NP>(r1, voff, b1 + u * ab.sstep, rb[u]);
                }
        }
        uint32_t msf[XB], msb[XB];
#pragma unroll
        for (int i = 0; i < XB; ++i) msf[i] = msb[i] = 0;
#pragma unroll
        for (int u = 0; u < XB; ++u)
            if (u < rem) {
                const int t = t0 + u;
                if (u == 0 && G > 0) sf.store_normalised(efrow + (size_t)G * vec);
                if (u == cb && t > 0) sb.store_normalised(ebrow + (size_t)((n - 1 - t) / XB) * vec);
                us2 Lf[NP], Lb[NP];
                sgm_step_pair<NP>(sf, rf[u], Lf, sb, rb[u], Lb, P1v, P2);
                msf[u] = sf.m;
                // x = rem-1-u
#pragma unroll
                for (int i = 0; i < XB; ++i) msb[i] = i == rem - 1 - u ? sb.m : msb[i];
            }
        store_minima<XB>(mfrow + (size_t)G * (XB / 2), msf, lane);
        {
            uint32_t v = msb[0];
#pragma unroll
            for (int i = 1; i < XB; ++i) v = lane == i ? msb[i] : v;
            if (lane < rem) mbrow[lane] = (uint16_t)v;
        }
    }
}

struct RowSide {                     // what k_rowsweep left behind
    const uint32_t* entF; const uint32_t* entB;
    const uint16_t* MF; const uint16_t* MB;
    int nbx;
};

// ACC: S already holds another family's sum (S += ...); otherwise this kernel writes S first.
// ONE (5-path mode, MODE_SGBM): of the column family only the DOWNWARD path (path 2) exists.  The walk is the same -- the hand-over
// block is what the row phase works on, and S has to be touched anyway -- but the upward path's costs are left out of the sum:
// in the upper half of the image (walked downwards) path 2 is the forward recomputation and the backward path is dropped, in the
// lower half (walked upwards) path 2 is the backward path, arriving with the upper half's end state, and the forward
// recomputation's costs are dropped.  S += L_2 + L_0 + L_4.
template <int NP, int K, bool ACC, bool ONE>
__global__ void __launch_bounds__(64 * Fuse<NP>::XB) __attribute__((amdgpu_waves_per_eu(Fuse<NP>::MINW, Fuse<NP>::MAXW)))
k_pairx(const uint32_t* __restrict__ C, uint32_t* __restrict__ S, const uint32_t* __restrict__ ckpt, const uint16_t* __restrict__ mins,
        const RowSide rs, int width1, int h, int P1, int P2, int maxseg, const uint32_t* __restrict__ endstate)
{
    constexpr int XB = Fuse<NP>::XB;
    static_assert(K % 2 == 0 && K <= XB && XB % 2 == 0, "one wave per row of a K-row segment; minima records are read as dwords");
    static_assert((size_t)XB * 2 * K * 256 * NP <= 160 * 1024, "the hand-over block must fit a CU's LDS");
    constexpr int VW = 64 * NP, VB = 256 * NP;
    // hand-over slots [2][K][XB][VW]: [0] cost vectors, [1] forward path costs; slot (element of a K-row segment), column.
    // The column is the INNER index: a row-phase wave reaches all ten columns of its slot, and a column-phase wave both
    // kinds of a slot, with immediate offsets from one address register (with the column outermost the compiler kept
    // four address registers per wave alive across the loop, and spilled them).
    extern __shared__ __attribute__((aligned(16))) uint32_t hand_raw[];
    constexpr int SS = XB * VW;                            // slot stride, dwords
    constexpr int KS = K * XB * VW;                        // kind stride
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int bx = blockIdx.x >> 1, half = blockIdx.x & 1;
    // ONE: which of the two column paths of this half counts (wave-uniform masks; all ones otherwise)
    const uint32_t keepF = (!ONE || half == 0) ? 0xFFFFFFFFu : 0u, keepB = (!ONE || half == 1) ? 0xFFFFFFFFu : 0u;
    auto drop = [](us2 (&v)[NP], uint32_t keep) {
        if constexpr (ONE) {
#pragma unroll
            for (int j = 0; j < NP; ++j) v[j] = as_us2(as_u32(v[j]) & keep);
        }
    };
    const int X0 = bx * XB, ncol = min(XB, width1 - X0);
    const bool colact = wv < ncol;                         // this wave owns a column (the last block of a row may be short)
    const int x = colact ? X0 + wv : X0;
    const int c = 2 * x + half;                            // sub-chain index of the split column family (half_chain_geometry)
    int dx = 0, dy = 1, x0, y0, n;
    half_chain_geometry(c, dx, dy, width1, h, x0, y0, n);  // the same n, y0, dy for every wave of the workgroup
    if (n == 0) return;                                    // h == 1: the first half of every column is empty
    const long long vec = VW;
    ChainAddr a;
    a.pixstep = (long long)dy * width1;
    a.pix0 = (long long)y0 * width1 + x0;
    a.sstep = (int)a.pixstep * VB;
    const uint32_t voff = lane * NP * 4;
    const uint32_t bK = a.bias(K);
    const us2 P1v = pk_splat(P1);
    const int F = n / K, r = n - F * K;
    const int top = r > 0 ? F : F - 1;                     // the segment the forward pass starts with (n >= 1)
    const rsrc_t ckr = mk_rsrc(ckpt + (long long)c * maxseg * vec);
    const uint32_t* mrow = (const uint32_t*)(mins + (size_t)c * maxseg * K);
    uint32_t* hc = hand_raw + wv * VW + lane * NP;         // this wave's column: slot e at hc + e * SS, forward costs at + KS

    // ---- the row phase: wave e takes element e of the K-row segment seg, whose vectors lie in slot (rev ? K-1-e : e) of
    // every wave's hand-over region
    // what the row phase of a segment needs from HBM: requested before the column phase that precedes it
    us2 eF[NP], eB[NP];
    Rec<XB, (WASS_VREC & 4) != 0> mf, mb;                  // minima records of the row
    auto row_fetch = [&](int seg, int cnt) {
        if (wv < cnt) {
            const int yrow = y0 + (seg * K + wv) * dy;
            if (bx >= 1) buf_ld<NP>(mk_rsrc(rs.entF + ((size_t)yrow * rs.nbx + bx) * vec), voff, 0, eF);
            else {
#pragma unroll
                for (int j = 0; j < NP; ++j) eF[j] = pk_splat(0);
            }
            if (bx + 1 < rs.nbx) buf_ld<NP>(mk_rsrc(rs.entB + ((size_t)yrow * rs.nbx + bx) * vec), voff, 0, eB);
            else {
#pragma unroll
                for (int j = 0; j < NP; ++j) eB[j] = pk_splat(0);
            }
            const uint32_t* pf = (const uint32_t*)(rs.MF + ((size_t)yrow * rs.nbx + bx) * XB);
            const uint32_t* pb = (const uint32_t*)(rs.MB + ((size_t)yrow * rs.nbx + bx) * XB);
            mf.load(pf, lane);
            mb.load(pb, lane);
        }
    };
    auto row_phase = [&](int seg, int cnt, bool rev) {
        __syncthreads();
        if (wv < cnt) {
            const int e = wv, slot = rev ? K - 1 - e : e;
            auto mat = [](const Rec<XB, (WASS_VREC & 4) != 0>& m, int i) { return m.at(i); };
            // Cost vectors are fetched from the slots when a path gets to them (twice per column) instead of all at the
            // start, and a column's forward path costs are updated as soon as both row paths have been there: with the ten
            // cost vectors and ten partial sums all live the kernel spilled, and every reload of a spilled register is a
            // wait for ALL outstanding loads -- the ring refills of the column phase included.
            uint32_t* rb = hand_raw + slot * SS + lane * NP;       // column j of the slot: rb + j * VW, its forward costs + KS
            PathState<NP> fa, fb;
            fa.load_normalised(eF);
            fb.load_normalised(eB);
            auto add_into_lf = [&](int j, const us2 (&x)[NP]) {
                uint32_t* lp = rb + KS + j * VW;
                us2 t[NP];
                lds_ld<NP>(lp, t);
#pragma unroll
                for (int q = 0; q < NP; ++q) t[q] = pk_adds(t[q], x[q]);
                lds_st<NP>(lp, t);
            };
            if (ncol == XB) {
                us2 cfn[NP], cbn[NP];
                lds_ld<NP>(rb, cfn);
                lds_ld<NP>(rb + (XB - 1) * VW, cbn);
#pragma unroll
                for (int u = 0; u < XB; ++u) {
                    const int jf = u, jb = XB - 1 - u;
                    us2 ca[NP], cb[NP], Lf[NP], Lb[NP];
#pragma unroll
                    for (int j = 0; j < NP; ++j) { ca[j] = cfn[j]; cb[j] = cbn[j]; }
                    if (u + 1 < XB) {
                        lds_ld<NP>(rb + (jf + 1) * VW, cfn);
                        lds_ld<NP>(rb + (jb - 1) * VW, cbn);
                    }
                    // minimum before the step: what the sweep recorded after the previous pixel of the path (0 behind an entry state)
                    sgm_step_ff<NP>(fa, u == 0 ? 0u : mat(mf, jf - 1), ca, Lf, fb, u == 0 ? 0u : mat(mb, jb + 1), cb, Lb, P1v, P2);
                    add_into_lf(jf, Lf);
                    add_into_lf(jb, Lb);
                }
            } else {                                        // the short block at the right image border
                for (int j = 0; j < ncol; ++j) {
                    us2 cv[NP], L[NP];
                    lds_ld<NP>(rb + j * VW, cv);
                    sgm_step_f<NP>(fa, j == 0 ? 0u : mf.at_dyn(j - 1), cv, L, P1v, P2);
                    add_into_lf(j, L);
                }
                for (int j = ncol - 1; j >= 0; --j) {
                    us2 cv[NP], L[NP];
                    lds_ld<NP>(rb + j * VW, cv);
                    sgm_step_f<NP>(fb, j == ncol - 1 ? 0u : mb.at_dyn(j + 1), cv, L, P1v, P2);
                    add_into_lf(j, L);
                }
            }
        }
        __syncthreads();
    };

    PathState<NP> bw, fw;
    us2 cf[K][NP];                                         // ring: cost vectors of the segment the forward recomputation covers next
    us2 sr[ACC ? K : 1][NP];                               // ring (ACC): S of the segment the backward path covers next
    us2 nvB[NP];
    Rec<K, (WASS_VREC & 2) != 0> mB;
    mB.clear();
#pragma unroll
    for (int j = 0; j < NP; ++j) nvB[j] = pk_splat(0);
    // ---- iteration 0: the forward path over segment `top` (with its reductions: once per chain), natural slot order.
    // What the LOOP consumes first (checkpoint and minima record of the next segment) is requested first: vmcnt counts in
    // order, and the waits inside the loop are computed for the worst of its predecessors.
    {
        const int cn = top == F ? r : K;
        if (colact) {
            if (top >= 2) buf_ld<NP>(ckr, voff, (uint32_t)(top - 2) * VB, nvB);
            if (top >= 1) mB.load(mrow + (size_t)(top - 1) * (K / 2), lane);
        }
        row_fetch(top, cn);
        if (colact) {
            bw.reset();
            {                                              // the backward path arrives from the other half of the chain
                us2 nv[NP];
                ld_stream_vec<NP>(endstate + (long long)(c ^ 1) * vec + lane * NP, nv);
                bw.load_normalised(nv);
            }
            const rsrc_t rc = a.run<NP>(C, (long long)top * K, cn);
            const uint32_t bc = a.bias(cn);
            us2 c0[K][NP];
#pragma unroll
            for (int u = 0; u < K; ++u)
                if (u < cn) buf_ld<NP>(rc, voff, bc + u * a.sstep, c0[u]);
            if constexpr (ACC) {
                const rsrc_t rs0 = a.run<NP>(S, (long long)top * K, cn);
#pragma unroll
                for (int u = 0; u < K; ++u)
                    if (u < cn) buf_ld<NP>(rs0, voff, bc + u * a.sstep, sr[u]);
            }
            fw.reset();
            if (top >= 1) {
                us2 nv[NP];
                buf_ld<NP>(ckr, voff, (uint32_t)(top - 1) * VB, nv);
                fw.load_normalised(nv);
                const rsrc_t rc2 = a.run<NP>(C, (long long)(top - 1) * K, K);
#pragma unroll
                for (int u = 0; u < K; ++u) buf_ld<NP>(rc2, voff, bK + u * a.sstep, cf[u]);
            }
#pragma unroll
            for (int u = 0; u < K; ++u)
                if (u < cn) {
                    us2 L[NP];
                    sgm_step<NP>(fw, c0[u], L, P1v, P2);
                    drop(L, keepF);
                    lds_st<NP>(hc + u * SS, c0[u]);
                    lds_st<NP>(hc + KS + u * SS, L);
                }
        }
        // Once per chain: nothing is in flight when the loop is entered, so that the waits inside it are the ones its own
        // back-edge needs (they are computed for the worst predecessor; with guarded loads pending here that was "wait
        // for everything" at the top of every row phase and every column phase).
        __builtin_amdgcn_s_waitcnt(0x0F70);                // vmcnt(0), expcnt and lgkmcnt untouched
    }

    // ---- iterations t = 1, 2, ...: backward path over segment s = top, top-1, ..., 0 || forward recomputation of segment
    // s-1 || refill of the ring with segment s-2.  NAT_: the slots hold segment s in natural order (element e in slot e); the
    // forward results go into the slots as they are drained, i.e. in reversed order, and so on alternately.
    // The common case, complete segments on both sides:
#define WASS_PX_FAST(s_)                                                                                               \
    if (colact) {                                                                                                       \
        const rsrc_t rcN = a.run<NP>(C, (long long)max((s_) - 2, 0) * K, K);                                              \
        const rsrc_t rsO = a.run<NP>(S, (long long)(s_) * K, K);                                                          \
        const rsrc_t rsN = a.run<NP>(S, (long long)((s_) - 1) * K, K);                                                    \
        us2 nvC[NP];                                                                                                    \
        Rec<K, (WASS_VREC & 2) != 0> mC;                                                                                                    \
        if ((s_) >= 3) buf_ld<NP>(ckr, voff, (uint32_t)((s_) - 3) * VB, nvC);                                            \
        else {                                                                                                          \
            _Pragma("unroll") for (int j = 0; j < NP; ++j) nvC[j] = pk_splat(0);                                         \
        }                                                                                                               \
        mC.load(mrow + (size_t)max((s_) - 2, 0) * (K / 2), lane);                                                \
        fw.load_normalised(nvB);                                                                                        \
        const int stp = nat ? -SS : SS;            /* the slots are drained from the end they were filled last */      \
        uint32_t* pc = hc + (nat ? (K - 1) * SS : 0);                                                                   \
        us2 cbn[NP], lfn[NP];                                                                                           \
        lds_ld<NP>(pc, cbn);                                                                                            \
        lds_ld<NP>(pc + KS, lfn);                                                                                       \
        _Pragma("unroll") for (int u = 0; u < K; ++u) {                                                                 \
            const int v = K - 1 - u;                                                                                    \
            uint32_t* pn = pc + stp;                                                                                    \
            us2 cb[NP], lfv[NP], Lf[NP], Lb[NP], sv[NP];                                                                \
            _Pragma("unroll") for (int j = 0; j < NP; ++j) { cb[j] = cbn[j]; lfv[j] = lfn[j]; }                          \
            if (u + 1 < K) {                                                                                            \
                lds_ld<NP>(pn, cbn);                                                                                    \
                lds_ld<NP>(pn + KS, lfn);                                                                               \
            }                                                                                                           \
            sgm_step_fb<NP>(fw, mB.before(u), cf[u], Lf, bw, cb, Lb, P1v, P2);                                   \
            drop(Lf, keepF);                                                                                            \
            drop(Lb, keepB);                                                                                            \
            _Pragma("unroll") for (int j = 0; j < NP; ++j) sv[j] = pk_adds(lfv[j], Lb[j]);                               \
            if constexpr (ACC) { _Pragma("unroll") for (int j = 0; j < NP; ++j) sv[j] = pk_adds(sv[j], sr[v][j]); }     \
            buf_st<NP>(rsO, voff, bK + v * a.sstep, sv);                                                                \
            lds_st<NP>(pc, cf[u]);                                                                                      \
            lds_st<NP>(pc + KS, Lf);                                                                                    \
            buf_ld<NP>(rcN, voff, bK + u * a.sstep, cf[u]);                                                              \
            if constexpr (ACC) buf_ld<NP>(rsN, voff, bK + v * a.sstep, sr[v]);                                           \
            pc = pn;                                                                                                    \
        }                                                                                                               \
        _Pragma("unroll") for (int j = 0; j < NP; ++j) nvB[j] = nvC[j];                                                 \
        mB = mC;                                                                                                        \
    }
    // Guarded form: the backward path over the short tail segment (cb_ < K elements) and/or no segment left to recompute
#define WASS_PX_SLOW(nat_, s_, cb_, hasfw_)                                                                             \
    if (colact) {                                                                                                       \
        const rsrc_t rcN = a.run<NP>(C, (long long)max((s_) - 2, 0) * K, K);                                              \
        const rsrc_t rsO = a.run<NP>(S, (long long)(s_) * K, (cb_));                                                      \
        const uint32_t bO = a.bias(cb_);                                                                                \
        const rsrc_t rsN = a.run<NP>(S, (long long)max((s_) - 1, 0) * K, K);                                              \
        us2 nvC[NP];                                                                                                    \
        Rec<K, (WASS_VREC & 2) != 0> mC;                                                                                                    \
        if ((s_) >= 3) buf_ld<NP>(ckr, voff, (uint32_t)((s_) - 3) * VB, nvC);                                            \
        else {                                                                                                          \
            _Pragma("unroll") for (int j = 0; j < NP; ++j) nvC[j] = pk_splat(0);                                         \
        }                                                                                                               \
        mC.load(mrow + (size_t)max((s_) - 2, 0) * (K / 2), lane);                                                \
        fw.load_normalised(nvB);                                                                                        \
        _Pragma("unroll") for (int u = 0; u < K; ++u) {                                                                 \
            const int v = K - 1 - u;                                                                                    \
            const int slot = (nat_) ? v : u;                                                                             \
            us2 Lf[NP];                                                                                                 \
            if (v < (cb_)) {                                                                                            \
                us2 cb[NP], lfv[NP], Lb[NP], sv[NP];                                                                    \
                lds_ld<NP>(hc + slot * SS, cb);                                                                         \
                lds_ld<NP>(hc + KS + slot * SS, lfv);                                                                   \
                sgm_step<NP>(bw, cb, Lb, P1v, P2);                                                                      \
                drop(Lb, keepB);                                                                                        \
                _Pragma("unroll") for (int j = 0; j < NP; ++j) sv[j] = pk_adds(lfv[j], Lb[j]);                           \
                if constexpr (ACC) { _Pragma("unroll") for (int j = 0; j < NP; ++j) sv[j] = pk_adds(sv[j], sr[v][j]); } \
                buf_st<NP>(rsO, voff, bO + v * a.sstep, sv);                                                            \
            }                                                                                                           \
            if (hasfw_) {                                                                                               \
                if constexpr (ACC) buf_ld<NP>(rsN, voff, bK + v * a.sstep, sr[v]);   /* segment s-1 is complete */       \
                sgm_step_f<NP>(fw, mB.before(u), cf[u], Lf, P1v, P2);                                            \
                drop(Lf, keepF);                                                                                        \
                lds_st<NP>(hc + slot * SS, cf[u]);                                                                      \
                lds_st<NP>(hc + KS + slot * SS, Lf);                                                                    \
                buf_ld<NP>(rcN, voff, bK + u * a.sstep, cf[u]);                                                          \
            }                                                                                                           \
        }                                                                                                               \
        _Pragma("unroll") for (int j = 0; j < NP; ++j) nvB[j] = nvC[j];                                                 \
        mB = mC;                                                                                                        \
    }
    // The hot loop holds ONLY the unguarded form.  With both forms in one loop the register allocator gave the ring
    // different registers in each and copied at the join -- behind an s_waitcnt vmcnt(0), i.e. every refill requested
    // during a column phase was waited for at its end and the row records requested before it at its start (cycle
    // counters: 12.3k cycles per column phase in the eight waves with a row role, 6.0k in the two without).
    int s = top;
    bool nat = true;                                       // iteration 0 filled the slots in natural order
    if (top == F) {                                        // r > 0: the backward path starts on the short tail segment
        row_phase(s, r, !nat);
        const bool hasfw = s >= 1;
        if (hasfw) row_fetch(s - 1, K);
        WASS_PX_SLOW(nat, s, r, hasfw)
        if (!hasfw) return;
        nat = !nat;                                        // the forward results went into the slots as they were drained
        --s;
        // Everything this guarded form requested has landed before the loop is entered: the waits inside the loop are
        // computed for the worst of its predecessors, and here that is "no refill was issued after the loads the loop
        // consumes first" -- which put a vmcnt(0) at the top of every column phase.
        __builtin_amdgcn_s_waitcnt(0x0F70);                // vmcnt(0), expcnt and lgkmcnt untouched
    }
    while (s >= 1) {
        row_phase(s, K, !nat);
        row_fetch(s - 1, K);
        WASS_PX_FAST(s)
        nat = !nat;
        --s;
    }
    row_phase(0, K, !nat);
    WASS_PX_SLOW(nat, 0, K, false)
#undef WASS_PX_FAST
#undef WASS_PX_SLOW
}

CkptLayout ckpt_layout(const SgmDims& d)
{
    CkptLayout L;
    L.K = ckpt_k(d.NP);
    // the column family's forward sweeps ride on the cost stage (k_vsum_col) whenever a pair kernel of that family follows: always
    // with 8 paths, and with 5 paths where the fused kernel is built (the column walk then carries path 2 and both row paths)
    const bool fused = d.NP <= WASS_FUSE_NP;
    L.cols_from_cost = d.ndirs == 8 || fused;
    auto add = [&](int dx, int dy, int smode) {
        const int f = L.nfam++;
        L.dx[f] = dx; L.dy[f] = dy; L.smode[f] = smode;
        L.nch[f] = dy == 0 ? d.h : (dx == 0 ? d.width1 : d.width1 + d.h - 1);
        int maxlen = dy == 0 ? d.width1 : (dx == 0 ? d.h : (d.width1 < d.h ? d.width1 : d.h));
        // Every family is split in the middle (half_chain_geometry) for twice the waves: 2 058 rows / 2 456 columns are
        // two waves per SIMD, and the longest chain of a diagonal family halves.  Measured at config B, same box, three
        // runs each: columns split 1.40 -> 1.11 ms (k_vsum_col), diagonals split 7.90 -> 7.60 ms (aggregation).
        // The column family is only split when the cost stage produces it.
        L.split[f] = dx != 0 || L.cols_from_cost;
        if (L.split[f]) { L.nch[f] *= 2; maxlen = maxlen - maxlen / 2; }
        L.mseg[f] = (maxlen + L.K - 1) / L.K;
        const size_t b = (size_t)L.nch[f] * (L.mseg[f] + (L.split[f] ? 1 : 0)) * (64 * d.NP) * sizeof(uint32_t);   // + the end states
        L.off[f + 1] = L.off[f] + ((b + 255) & ~(size_t)255);
    };
    // The kernel that carries the winner-take-all goes last and should have the most chains (the WTA adds ~50
    // instructions per pixel): the anti-diagonals (width1 + h - 1 chains).
    if (d.ndirs == 8 && fused) {
        L.rows_fused = true;
        // Order of the pair kernels: diagonals, columns + rows, anti-diagonals.  The fused kernel needs the row sweeps, whose
        // chains are the longest in the image (1.4 ms alone); with the diagonal family first the main stream has 2 ms of
        // its own work to do while they run (launch_aggregate_np), and only one checkpoint sweep is left to run beside
        // the fused kernel instead of two.  S: written by the diagonals, read-modify-written by k_pairx, read by the last.
        add(0, 1, 1);                // columns + rows: paths 2 + 6 and 0 + 4 (k_pairx, S +=), second
        add(1, 1, 0);                // diagonals:      paths 1 + 7 (S written), first
        add(-1, 1, 2);               // anti-diagonals: paths 3 + 5, winner-take-all fused, last
    } else if (d.ndirs == 8) {
        // D > 512: a pixel vector is 1.25 KiB or more; the fused kernel's hand-over block (XB columns x K rows, two kinds) does not
        // fit a CU's LDS with K >= 4 and >= 8 columns.  One pair kernel per family.
        add(0, 1, 0);                // columns:        paths 2 + 6   (S written)
        add(1, 0, 1);                // rows:           paths 0 + 4
        add(1, 1, 1);                // diagonals:      paths 1 + 7
        add(-1, 1, 2);               // anti-diagonals: paths 3 + 5, winner-take-all fused
    } else if (fused) {
        // 5 paths (MODE_SGBM, what the reference runs), round 6: paths 1, 2 and 3 have no partner, so each used to be a pass of its
        // own over S (S = L_2 written by the cost stage, += L_1, += L_0 + L_4, + L_3 and selection: S written three times, read three
        // times).  Now path 1's sweep writes S first (beside the row sweeps on the side stream), the fused kernel of the 8-path
        // schedule adds path 2 and both row paths in one pass (k_pairx<.., ONE>), path 3's sweep reads S and selects: S written
        // twice and read twice, and the cost stage carries only the column family's checkpoints.
        L.rows_fused = true;
        add(0, 1, 1);                // column walk: path 2 and rows 0 + 4 (k_pairx<.., ONE>, S +=)
    } else {
        L.path2_from_cost = true;
        add(1, 0, 1);                // rows: paths 0 + 4, added to the S = L_2 that the cost stage left behind
    }
    size_t o = L.off[L.nfam];
    for (int f = 0; f < L.nfam; ++f) {
        L.moff[f] = o;
        o += ((size_t)L.nch[f] * L.mseg[f] * L.K * sizeof(uint16_t) + 255) & ~(size_t)255;
    }
    if (L.rows_fused) {              // k_rowsweep: entry states per block of XB columns and minima per pixel, both row paths
        const int XB = fuse_xb(d.NP);
        L.nbx = (d.width1 + XB - 1) / XB;
        const size_t eb = ((size_t)d.h * L.nbx * (64 * d.NP) * sizeof(uint32_t) + 255) & ~(size_t)255;
        const size_t mb = ((size_t)d.h * L.nbx * XB * sizeof(uint16_t) + 255) & ~(size_t)255;
        L.roff[0] = o; o += eb;
        L.roff[1] = o; o += eb;
        L.roff[2] = o; o += mb;
        L.roff[3] = o; o += mb;
    }
    L.total = o;
    return L;
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
    auto nchains = [&](int dx, int dy) { return dy == 0 ? d.h : (dx == 0 ? d.width1 : d.width1 + d.h - 1); };
    // every family has its own checkpoint region so that all checkpoint sweeps (which only read C) can run ahead
    // on the side stream while the main stream accumulates S
    const CkptLayout lay = ckpt_layout(d);
    const int nf = lay.nfam;
    int rc = ensure(c, c->ckpt, lay.total);
    if (rc) return rc;
    KernelClock kc(c);                               // per-kernel hipEvents, only when the context asks for them (wass_ctx_set_kernel_events)

    WASS_HIP(c, hipEventRecord(c->ev_cost, c->stream));
    WASS_HIP(c, hipStreamWaitEvent(c->side, c->ev_cost, 0));
    char* const ckb = (char*)c->ckpt.p;
    if (lay.rows_fused) {                            // the row sweeps come first: the first kernel on the main stream needs them
        if constexpr (NP <= WASS_FUSE_NP) {
            kc.begin("k_rowsweep", c->side);
            hipLaunchKernelGGL((k_rowsweep<NP>), dim3((d.h + 3) / 4), dim3(256), 0, c->side, C, (uint32_t*)(ckb + lay.roff[0]),
                               (uint32_t*)(ckb + lay.roff[1]), (uint16_t*)(ckb + lay.roff[2]), (uint16_t*)(ckb + lay.roff[3]), d.width1, d.h,
                               d.P1, d.P2, lay.nbx);
            kc.end(c->side);
            WASS_HIP(c, hipEventRecord(c->ev_ckpt[3], c->side));
            ++nl;
        }
    }
    const bool three = lay.rows_fused && nf == 3;    // the 8-path schedule with the fused kernel in the middle
    for (int f = 0; f < nf; ++f) {
        if (f == 0 && lay.cols_from_cost) {         // written by k_vsum_col on the main stream already
            WASS_HIP(c, hipEventRecord(c->ev_ckpt[f], c->stream));
            continue;
        }
        const int nch = lay.nch[f], mseg = lay.mseg[f];
        uint32_t* ckf = (uint32_t*)((char*)c->ckpt.p + lay.off[f]);
        // (The sweeps that run BESIDE the row sweeps stretch them -- 2 455 dependent steps per chain -- from 1.4 to 2.6 ms.
        // That was a loss while the fused kernel, which needs them, came first; it is free now that the diagonal family's
        // sweep and pair kernel, 2 ms of main-stream work, come first.  s_setprio in k_rowsweep changes nothing.)
        // fused schedule: the diagonal sweep goes on the MAIN stream (its pair kernel follows it there), beside the row sweeps
        hipStream_t ss = (three && f == 1) ? c->stream : c->side;
        kc.begin(f == 1 ? "k_ckpt(family 1)" : (f == 2 ? "k_ckpt(family 2)" : (f == 3 ? "k_ckpt(family 3)" : "k_ckpt(family 0)")), ss);
        hipLaunchKernelGGL((k_ckpt<NP, K>), dim3((nch + 3) / 4), dim3(256), 0, ss, C, ckf, (uint16_t*)((char*)c->ckpt.p + lay.moff[f]),
                           d.width1, d.h, lay.dx[f], lay.dy[f], d.P1, d.P2, nch, mseg,
                           lay.split[f] ? ckf + (size_t)nch * mseg * (64 * NP) : (uint32_t*)nullptr);
        kc.end(ss);
        WASS_HIP(c, hipEventRecord(c->ev_ckpt[f], ss));
        ++nl;
    }

    if (d.ndirs == 5) {
        // path 1 needs no checkpoints: it runs while the row sweeps / row checkpoints are produced.  Fused schedule: it is the first
        // writer of S (SMODE 0); otherwise it adds to the S = L_2 of the cost stage.
        const int nch = nchains(1, 1);
        kc.begin("k_sweep(path 1)", c->stream);
        if (lay.rows_fused)
            hipLaunchKernelGGL((k_sweep<NP, 0, U>), dim3((nch + 3) / 4), dim3(256), 0, c->stream, C, S, d.width1, d.h, 1, 1, d.P1, d.P2,
                               nch, d.D, d.minD, d.uniq, c->debug ? 1 : 0, sd, sk);
        else
            hipLaunchKernelGGL((k_sweep<NP, 1, U>), dim3((nch + 3) / 4), dim3(256), 0, c->stream, C, S, d.width1, d.h, 1, 1, d.P1, d.P2,
                               nch, d.D, d.minD, d.uniq, c->debug ? 1 : 0, sd, sk);
        kc.end(c->stream);
        ++nl;
    }
    for (int fi = 0; fi < nf; ++fi) {
        const int fused_order[3] = { 1, 0, 2 };
        const int f = three ? fused_order[fi] : fi;
        const int nch = lay.nch[f], mseg = lay.mseg[f];
        const uint32_t* ck = (const uint32_t*)((char*)c->ckpt.p + lay.off[f]);
        const uint16_t* mn = (const uint16_t*)((char*)c->ckpt.p + lay.moff[f]);
        const dim3 grid((nch + 3) / 4), block(256);
        WASS_HIP(c, hipStreamWaitEvent(c->stream, c->ev_ckpt[f], 0));
#define WASS_PAIR(SMODE)                                                                                         \
        hipLaunchKernelGGL((k_pair<NP, K, SMODE>), grid, block, 0, c->stream, C, S, ck, mn, d.width1, d.h, lay.dx[f],  \
                           lay.dy[f], d.P1, d.P2, nch, mseg, d.D, d.minD, d.uniq, c->debug ? 1 : 0, sd, sk,            \
                           lay.split[f] ? (const uint32_t*)(ck + (size_t)nch * mseg * (64 * NP)) : (const uint32_t*)nullptr)
        if (f == 0 && lay.rows_fused) {
            // only the instances that can be launched are instantiated (the others would need more LDS than a CU has)
            if constexpr (NP <= WASS_FUSE_NP) {
                WASS_HIP(c, hipStreamWaitEvent(c->stream, c->ev_ckpt[3], 0));
                const RowSide rs = { (const uint32_t*)(ckb + lay.roff[0]), (const uint32_t*)(ckb + lay.roff[1]),
                                     (const uint16_t*)(ckb + lay.roff[2]), (const uint16_t*)(ckb + lay.roff[3]), lay.nbx };
                constexpr int XB = Fuse<NP>::XB;
                const size_t ldsx = (size_t)XB * 2 * K * (64 * NP) * sizeof(uint32_t);
                kc.begin("k_pairx", c->stream);
                if (d.ndirs == 5) {
                    WASS_HIP(c, hipFuncSetAttribute((const void*)k_pairx<NP, K, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsx));
                    hipLaunchKernelGGL((k_pairx<NP, K, true, true>), dim3(2 * lay.nbx), dim3(64 * XB), ldsx, c->stream, C, S, ck, mn, rs, d.width1, d.h,
                                       d.P1, d.P2, mseg, (const uint32_t*)(ck + (size_t)nch * mseg * (64 * NP)));
                } else {
                    WASS_HIP(c, hipFuncSetAttribute((const void*)k_pairx<NP, K, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsx));
                    hipLaunchKernelGGL((k_pairx<NP, K, true, false>), dim3(2 * lay.nbx), dim3(64 * XB), ldsx, c->stream, C, S, ck, mn, rs, d.width1, d.h,
                                       d.P1, d.P2, mseg, (const uint32_t*)(ck + (size_t)nch * mseg * (64 * NP)));
                }
                kc.end(c->stream);
            } else {
                return set_err(c, WASS_ERR_UNSUPPORTED, "row fusion is not built for NP = %d", NP);
            }
        } else {
            kc.begin(f == 1 ? "k_pair(family 1)" : (f == 2 ? "k_pair(family 2)" : (f == 3 ? "k_pair(family 3)" : "k_pair(family 0)")), c->stream);
            if (lay.smode[f] == 0) WASS_PAIR(0);
            else if (lay.smode[f] == 1) WASS_PAIR(1);
            else WASS_PAIR(2);
            kc.end(c->stream);
        }
#undef WASS_PAIR
        ++nl;
    }
    if (d.ndirs == 5) {              // path 3, winner-take-all fused (paths 2, 1, 0 + 4 are in S by now)
        const int nch = nchains(-1, 1);
        kc.begin("k_sweep(path 3 + selection)", c->stream);
        hipLaunchKernelGGL((k_sweep<NP, 2, U>), dim3((nch + 3) / 4), dim3(256), 0, c->stream, C, S, d.width1, d.h, -1, 1, d.P1, d.P2,
                           nch, d.D, d.minD, d.uniq, c->debug ? 1 : 0, sd, sk);
        kc.end(c->stream);
        ++nl;
    }
    if (n_launches) *n_launches = nl;
    WASS_HIP(c, hipGetLastError());
    return WASS_OK;
}

// ---------------------------------------------------------------------------
// Device-side canary (wass_sgm_selftest): the aggregated volume S of the production schedule -- checkpoint sweeps, pair
// kernels with recomputation and LDS hand-over, row fusion -- against S built the plain way, one k_sweep per path over the
// same C.  The two share the arithmetic of one step (sgm_step) and nothing of the machinery around it, which is where a
// miscompiled loop, a late store or a missing wait would sit.  No CPU oracle involved: runs wherever the library runs.
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_count_diff(const uint32_t* __restrict__ A, const uint32_t* __restrict__ B, size_t nvec, int vecdw,
                                                    int D, unsigned long long* __restrict__ out)
{
    // one thread per dword; disparities d >= D are padding (0xFFFF costs) and are not compared
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    unsigned int bad = 0;
    if (i < nvec * (size_t)vecdw) {
        const int dw = (int)(i % (size_t)vecdw);
        const uint32_t a = A[i], b = B[i];
        if (2 * dw < D && (a & 0xFFFFu) != (b & 0xFFFFu)) ++bad;
        if (2 * dw + 1 < D && (a >> 16) != (b >> 16)) ++bad;
    }
    for (int o = 32; o > 0; o >>= 1) bad += __shfl_down(bad, o);
    if ((threadIdx.x & 63) == 0 && bad) atomicAdd(out, (unsigned long long)bad);
}

template <int NP>
static int selftest_reference_np(wass_ctx* c, const SgmDims& d, uint32_t* S2, unsigned long long* d_count)
{
    constexpr int U = ckpt_k(NP);
    const uint32_t* C = (const uint32_t*)c->C.p;
    // direction of travel = -r of Appendix A.4: paths 0..4 of MODE_SGBM, then 5..7 of MODE_HH
    static const int dirs[8][2] = { { 1, 0 }, { 1, 1 }, { 0, 1 }, { -1, 1 }, { -1, 0 }, { 1, -1 }, { 0, -1 }, { -1, -1 } };
    for (int r = 0; r < d.ndirs; ++r) {
        const int dx = dirs[r][0], dy = dirs[r][1];
        const int nch = dy == 0 ? d.h : (dx == 0 ? d.width1 : d.width1 + d.h - 1);
        const dim3 grid((nch + 3) / 4), block(256);
        if (r == 0)
            hipLaunchKernelGGL((k_sweep<NP, 0, U>), grid, block, 0, c->stream, C, S2, d.width1, d.h, dx, dy, d.P1, d.P2, nch, d.D, d.minD, d.uniq, 1,
                               (int16_t*)nullptr, (uint32_t*)nullptr);
        else
            hipLaunchKernelGGL((k_sweep<NP, 1, U>), grid, block, 0, c->stream, C, S2, d.width1, d.h, dx, dy, d.P1, d.P2, nch, d.D, d.minD, d.uniq, 1,
                               (int16_t*)nullptr, (uint32_t*)nullptr);
    }
    const size_t nvec = (size_t)d.h * d.width1;
    const size_t ndw = nvec * (size_t)(64 * NP);
    hipLaunchKernelGGL(k_count_diff, dim3((unsigned)((ndw + 255) / 256)), dim3(256), 0, c->stream, (const uint32_t*)c->S.p, (const uint32_t*)S2, nvec,
                       64 * NP, d.D, d_count);
    WASS_HIP(c, hipGetLastError());
    return WASS_OK;
}

int selftest_reference(wass_ctx* c, const SgmDims& d, uint32_t* S2, unsigned long long* d_count)
{
    switch (d.NP) {
        case 1: return selftest_reference_np<1>(c, d, S2, d_count);
        case 2: return selftest_reference_np<2>(c, d, S2, d_count);
        case 3: return selftest_reference_np<3>(c, d, S2, d_count);
        case 4: return selftest_reference_np<4>(c, d, S2, d_count);
        case 5: return selftest_reference_np<5>(c, d, S2, d_count);
        case 6: return selftest_reference_np<6>(c, d, S2, d_count);
        case 7: return selftest_reference_np<7>(c, d, S2, d_count);
        case 8: return selftest_reference_np<8>(c, d, S2, d_count);
    }
    return set_err(c, WASS_ERR_UNSUPPORTED, "MAX_DISPARITY %d not supported (max 1024)", d.D);
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
