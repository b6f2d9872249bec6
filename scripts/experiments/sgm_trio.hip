// sgm_trio.hip -- path aggregation by pipelined column strips (three paths, one pass, no recomputation).
//
// All paths whose predecessor lies in the previous column (or the same column) can be computed in ONE sweep
// that walks the image column strip by column strip:
//     xdir=+1, ydir=+1 :  r = (-1, 0), (-1,-1), ( 0,-1)   = paths 0, 1, 2 of SURVEY.md Appendix A.4
//     xdir=-1, ydir=-1 :  r = (+1, 0), (+1,+1), ( 0,+1)   = paths 4, 7, 6
//     xdir=-1, ydir=+1, no vertical path :  (+1, 0), (+1,-1) = paths 4, 3   (MODE_SGBM)
// One wavefront owns a strip of W columns for every row and keeps the three recurrences' states in
// registers; the only thing it needs from outside is the state of the neighbouring strip's last column
// (horizontal path: same row, diagonal path: previous row).  Strip j therefore runs at least one row behind
// strip j-1: a linear software pipeline across the wavefronts of one launch.
//
// Hand-off (MI355X_MICROARCH.md, "data IS the flag" form): the producer writes each boundary vector once with
// 8-byte agent-scope (sc1, write-through) stores; valid costs are < 0x8000, an unwritten slot holds 0xFF bytes.
// The consumer loads with agent-scope loads and re-polls until every lane sees a valid granule -- no flag, no
// fence, no s_waitcnt on the producer -- then writes the 0xFF pattern back so the buffer is ready for the next
// frame.  Loads for row t+1 are issued during row t, so a consumer that lags (the steady state) never waits.
// Every spin is bounded; on time-out the wave raises flags[0] bit 1 and leaves (the host then re-initialises
// the buffer and reports an error) -- no hang.  Forward progress is NOT guaranteed by residency: a launch has up to
// 1228 one-wave workgroups and the kernel runs about one wave per SIMD (1024 slots, fewer when other streams hold
// some), so it relies on workgroups being dispatched in ascending order -- a producer strip is then always resident or
// finished before its consumer spins.  That order is what the hardware does but no API promises it; if it ever failed,
// consumers would run into the spin bound and the frame would be reported as failed (never a wrong result, never a
// hang).  This schedule is an opt-in experiment (WASS_AGG=trio), not the production path.
//
// Compared with the chain kernels (sgm_aggregate.hip) this needs no checkpoint sweep and no forward
// recomputation: per cell, C is read once and S written once for three paths.
#include "sgm_step.h"

#include <stdlib.h>
#include <type_traits>

namespace wass {

// three independent recurrences on the same cost vector, advanced statement by statement
template <int NP, bool HAS_C>
__device__ __forceinline__ void sgm_step3(PathState<NP>& a, PathState<NP>& b, PathState<NP>& c, const us2 (&cv)[NP],
                                          us2 (&La)[NP], us2 (&Lb)[NP], us2 (&Lc)[NP], const us2 P1v, const uint32_t P2)
{
    a.shr = dpp_mov<DPP_WAVE_SHR1>(a.shr, as_u32(a.L[NP - 1]));
    b.shr = dpp_mov<DPP_WAVE_SHR1>(b.shr, as_u32(b.L[NP - 1]));
    if (HAS_C) c.shr = dpp_mov<DPP_WAVE_SHR1>(c.shr, as_u32(c.L[NP - 1]));
    a.shl = dpp_mov<DPP_WAVE_SHL1>(a.shl, as_u32(a.L[0]));
    b.shl = dpp_mov<DPP_WAVE_SHL1>(b.shl, as_u32(b.L[0]));
    if (HAS_C) c.shl = dpp_mov<DPP_WAVE_SHL1>(c.shl, as_u32(c.L[0]));
    const us2 amv = pk_splat(a.m), amp2 = pk_splat(a.m + P2);
    const us2 bmv = pk_splat(b.m), bmp2 = pk_splat(b.m + P2);
    const us2 cmv = pk_splat(c.m), cmp2 = pk_splat(c.m + P2);
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        const uint32_t alo = j == 0 ? a.shr : as_u32(a.L[j - 1]), ahi = j == NP - 1 ? a.shl : as_u32(a.L[j + 1]);
        const uint32_t blo = j == 0 ? b.shr : as_u32(b.L[j - 1]), bhi = j == NP - 1 ? b.shl : as_u32(b.L[j + 1]);
        us2 ax = pk_min(as_us2(__builtin_amdgcn_alignbit(as_u32(a.L[j]), alo, 16)), as_us2(__builtin_amdgcn_alignbit(ahi, as_u32(a.L[j]), 16)));
        us2 bx = pk_min(as_us2(__builtin_amdgcn_alignbit(as_u32(b.L[j]), blo, 16)), as_us2(__builtin_amdgcn_alignbit(bhi, as_u32(b.L[j]), 16)));
        ax = pk_adds(ax, P1v); bx = pk_adds(bx, P1v);
        ax = pk_min(a.L[j], ax); bx = pk_min(b.L[j], bx);
        ax = pk_min(ax, amp2); bx = pk_min(bx, bmp2);
        ax = ax - amv; bx = bx - bmv;
        La[j] = pk_adds(cv[j], ax); Lb[j] = pk_adds(cv[j], bx);
        if (HAS_C) {
            const uint32_t clo = j == 0 ? c.shr : as_u32(c.L[j - 1]), chi = j == NP - 1 ? c.shl : as_u32(c.L[j + 1]);
            us2 cx = pk_min(as_us2(__builtin_amdgcn_alignbit(as_u32(c.L[j]), clo, 16)), as_us2(__builtin_amdgcn_alignbit(chi, as_u32(c.L[j]), 16)));
            cx = pk_adds(cx, P1v);
            cx = pk_min(c.L[j], cx);
            cx = pk_min(cx, cmp2);
            cx = cx - cmv;
            Lc[j] = pk_adds(cv[j], cx);
        }
    }
    us2 am = La[0], bm = Lb[0], cm = HAS_C ? Lc[0] : pk_splat(0);
#pragma unroll
    for (int j = 1; j < NP; ++j) { am = pk_min(am, La[j]); bm = pk_min(bm, Lb[j]); if (HAS_C) cm = pk_min(cm, Lc[j]); }
    uint32_t ra = min((uint32_t)am.x, (uint32_t)am.y), rb = min((uint32_t)bm.x, (uint32_t)bm.y);
    uint32_t rc = min((uint32_t)cm.x, (uint32_t)cm.y);
    // three reductions in lock-step
    ra = min(ra, dpp_mov<DPP_QUAD_XOR1>(0xFFFFFFFFu, ra)); rb = min(rb, dpp_mov<DPP_QUAD_XOR1>(0xFFFFFFFFu, rb));
    if (HAS_C) rc = min(rc, dpp_mov<DPP_QUAD_XOR1>(0xFFFFFFFFu, rc));
    ra = min(ra, dpp_mov<DPP_QUAD_XOR2>(0xFFFFFFFFu, ra)); rb = min(rb, dpp_mov<DPP_QUAD_XOR2>(0xFFFFFFFFu, rb));
    if (HAS_C) rc = min(rc, dpp_mov<DPP_QUAD_XOR2>(0xFFFFFFFFu, rc));
    ra = min(ra, dpp_mov<DPP_ROW_HALF_MIRROR>(0xFFFFFFFFu, ra)); rb = min(rb, dpp_mov<DPP_ROW_HALF_MIRROR>(0xFFFFFFFFu, rb));
    if (HAS_C) rc = min(rc, dpp_mov<DPP_ROW_HALF_MIRROR>(0xFFFFFFFFu, rc));
    ra = min(ra, dpp_mov<DPP_ROW_MIRROR>(0xFFFFFFFFu, ra)); rb = min(rb, dpp_mov<DPP_ROW_MIRROR>(0xFFFFFFFFu, rb));
    if (HAS_C) rc = min(rc, dpp_mov<DPP_ROW_MIRROR>(0xFFFFFFFFu, rc));
    ra = min(ra, dpp_mov<DPP_ROW_BCAST15, 0xa>(0xFFFFFFFFu, ra)); rb = min(rb, dpp_mov<DPP_ROW_BCAST15, 0xa>(0xFFFFFFFFu, rb));
    if (HAS_C) rc = min(rc, dpp_mov<DPP_ROW_BCAST15, 0xa>(0xFFFFFFFFu, rc));
    ra = min(ra, dpp_mov<DPP_ROW_BCAST31, 0xc>(0xFFFFFFFFu, ra)); rb = min(rb, dpp_mov<DPP_ROW_BCAST31, 0xc>(0xFFFFFFFFu, rb));
    if (HAS_C) rc = min(rc, dpp_mov<DPP_ROW_BCAST31, 0xc>(0xFFFFFFFFu, rc));
    a.m = (uint32_t)__builtin_amdgcn_readlane((int)ra, 63);
    b.m = (uint32_t)__builtin_amdgcn_readlane((int)rb, 63);
    if (HAS_C) c.m = (uint32_t)__builtin_amdgcn_readlane((int)rc, 63);
#pragma unroll
    for (int j = 0; j < NP; ++j) { a.L[j] = La[j]; b.L[j] = Lb[j]; if (HAS_C) c.L[j] = Lc[j]; }
}

// ---- boundary vectors: NG = ceil(NP/2) 8-byte granules per lane -------------------------------------------
template <int NP> struct Halo { static constexpr int NG = (NP + 1) / 2; };
constexpr unsigned long long HALO_EMPTY = ~0ull;
typedef unsigned long long __attribute__((address_space(1))) gu64;

template <int NP>
__device__ __forceinline__ void halo_publish(unsigned long long* __restrict__ slot, int lane, const PathState<NP>& st)
{
    // normalised costs; padded disparity slots (0xFFFF) are clamped to 0x7FFF so that bit 15 can mark "empty":
    // a padded slot saturates again at the next step and 0x7FFF + P1 can never win a minimum against a real cost
    const us2 mv = pk_splat(st.m), cap = pk_splat(0x7FFF);
#pragma unroll
    for (int g = 0; g < Halo<NP>::NG; ++g) {
        const uint32_t lo = as_u32(pk_min(st.L[2 * g] - mv, cap));
        const uint32_t hi = 2 * g + 1 < NP ? as_u32(pk_min(st.L[2 * g + 1] - mv, cap)) : 0u;
        // write-through (sc1) 8-byte store: reaches memory without any fence, never torn
        __hip_atomic_store(slot + g * 64 + lane, ((unsigned long long)hi << 32) | lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
template <int NP>
__device__ __forceinline__ void halo_issue(const unsigned long long* __restrict__ slot, int lane, unsigned long long (&v)[Halo<NP>::NG])
{
#pragma unroll
    for (int g = 0; g < Halo<NP>::NG; ++g)
        v[g] = __hip_atomic_load(slot + g * 64 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// true when every lane holds a completely written vector (costs are < 0x8000; an empty slot is all ones)
template <int NP>
__device__ __forceinline__ bool halo_ready(const unsigned long long (&v)[Halo<NP>::NG])
{
    bool ok = true;
#pragma unroll
    for (int g = 0; g < Halo<NP>::NG; ++g) ok &= (v[g] & 0x8000800080008000ull) == 0;
    return __all(ok);
}
template <int NP>
__device__ __forceinline__ void halo_to_state(const unsigned long long (&v)[Halo<NP>::NG], PathState<NP>& st)
{
#pragma unroll
    for (int j = 0; j < NP; ++j) st.L[j] = as_us2((uint32_t)(v[j / 2] >> ((j & 1) * 32)));
    st.m = 0;
}
template <int NP>
__device__ __forceinline__ void halo_clear(unsigned long long* __restrict__ slot, int lane)
{
#pragma unroll
    for (int g = 0; g < Halo<NP>::NG; ++g)
        __hip_atomic_store(slot + g * 64 + lane, HALO_EMPTY, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// One launch = one pipelined sweep.  blockIdx.x = strip index in processing order.
// The row loop is unrolled by the prefetch ring depth R = 3: the cost vectors and boundary vectors of row t+2 are
// requested while row t computes, so the (in-order) wait for them is rarely behind still-outstanding older
// stores -- with a single wave per SIMD nothing else would hide that latency.
template <int NP, int W, bool HAS_V>
__global__ void __launch_bounds__(64) k_trio(const uint32_t* __restrict__ C, uint32_t* __restrict__ Sout,
                                             unsigned long long* halo, int width1, int h, int xdir, int ydir, int P1,
                                             int P2, int nstrips, uint32_t* __restrict__ flags)
{
    constexpr int NG = Halo<NP>::NG;
    constexpr int R = 3;                                                  // prefetch ring depth (rows)
    const int lane = threadIdx.x;
    const int j = blockIdx.x;
    const int xs = xdir > 0 ? j * W : width1 - 1 - j * W;                 // first column of the strip, in travel order
    const int ncols = min(W, width1 - j * W);
    const long long vec = 64 * NP;
    const size_t slot_sz = (size_t)NG * 64;                               // u64 per boundary vector
    // halo[strip][row t][2 vectors: horizontal, diagonal]
    unsigned long long* my_out = halo + (size_t)j * h * 2 * slot_sz;
    unsigned long long* my_in = halo + (size_t)(j > 0 ? j - 1 : 0) * h * 2 * slot_sz;
    const bool has_left = j > 0, has_right = j + 1 < nstrips;
    const us2 P1v = pk_splat(P1), cap = pk_splat(0x7FFF);

    PathState<NP> Ld[W], Lv[W], in_d;
    in_d.reset();
#pragma unroll
    for (int k = 0; k < W; ++k) { Ld[k].reset(); Lv[k].reset(); }

    auto cell = [&](int t, int k) -> long long {
        const int y = ydir > 0 ? t : h - 1 - t;
        return ((long long)y * width1 + (xs + k * xdir)) * vec + lane * NP;
    };
    us2 cring[R][W][NP];
    unsigned long long hring[R][2][NG];
    auto request = [&](int t, auto slot) {                                // issue every load row t needs
        constexpr int r = decltype(slot)::value;
#pragma unroll
        for (int k = 0; k < W; ++k)
            if (k < ncols) {
#pragma unroll
                for (int q = 0; q < NP; ++q) cring[r][k][q] = as_us2(C[cell(t, k) + q]);
            }
        if (has_left) {
            const unsigned long long* src = my_in + (size_t)t * 2 * slot_sz;
            halo_issue<NP>(src, lane, hring[r][0]);
            halo_issue<NP>(src + slot_sz, lane, hring[r][1]);
        }
    };
    bool dead = false;
    auto row = [&](int t, auto slot, auto slot2) {
        constexpr int r = decltype(slot)::value;
        if (t + 2 < h) request(t + 2, slot2);
        // ---- incoming state of the neighbouring strip's last column (this row)
        PathState<NP> Lh;
        Lh.reset();
        if (has_left) {
            unsigned long long* src = my_in + (size_t)t * 2 * slot_sz;
            unsigned spins = 0;
            while (!(halo_ready<NP>(hring[r][0]) && halo_ready<NP>(hring[r][1]))) {   // wave-uniform; rare once the pipe runs
                if (++spins > (1u << 22)) {                               // ~seconds: give up instead of hanging
                    if (lane == 0) atomicOr(flags, 2u);
                    dead = true;
                    return;
                }
                __builtin_amdgcn_s_sleep(1);
                halo_issue<NP>(src, lane, hring[r][0]);
                halo_issue<NP>(src + slot_sz, lane, hring[r][1]);
            }
            halo_to_state<NP>(hring[r][0], Lh);
            // leave the slots empty for the next frame (nobody else reads them)
            halo_clear<NP>(src, lane);
            halo_clear<NP>(src + slot_sz, lane);
        }
        // ---- the strip's cells of this row, in travel order
        PathState<NP> Ldn[W];
#pragma unroll
        for (int k = 0; k < W; ++k)
            if (k < ncols) {
                // diagonal predecessor: previous row, previous column (the neighbour strip for k = 0)
                PathState<NP> dstate = k == 0 ? in_d : Ld[k - 1];
                us2 La[NP], Lb[NP], Lc[NP];
                sgm_step3<NP, HAS_V>(Lh, dstate, Lv[k], cring[r][k], La, Lb, Lc, P1v, P2);
                Ldn[k] = dstate;
                uint32_t* so = Sout + cell(t, k);
#pragma unroll
                for (int q = 0; q < NP; ++q) {
                    us2 s = pk_adds(La[q], Lb[q]);
                    if (HAS_V) s = pk_adds(s, Lc[q]);
                    so[q] = as_u32(pk_min(s, cap));
                }
            }
        // ---- publish this strip's last column for the next strip (only full strips have one)
        if (has_right) {
            unsigned long long* dst = my_out + (size_t)t * 2 * slot_sz;
            halo_publish<NP>(dst, lane, Lh);
            halo_publish<NP>(dst + slot_sz, lane, Ldn[W - 1]);
        }
        // ---- the neighbour's diagonal state of THIS row feeds column 0 of the next row
        if (has_left) halo_to_state<NP>(hring[r][1], in_d);
#pragma unroll
        for (int k = 0; k < W; ++k)
            if (k < ncols) Ld[k] = Ldn[k];
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    request(0, I0{});
    if (h > 1) request(1, I1{});
    for (int t = 0; t < h; t += R) {
        row(t, I0{}, I2{});
        if (dead) return;
        if (t + 1 < h) row(t + 1, I1{}, I0{});
        if (dead) return;
        if (t + 2 < h) row(t + 2, I2{}, I1{});
        if (dead) return;
    }
}

// MODE_SGBM has no family left for a fused selection: S (paths 0,1,2) + S2 (paths 4,3) -> winner-take-all
template <int NP>
__global__ void __launch_bounds__(256) k_wta_sum(const uint32_t* __restrict__ S, const uint32_t* __restrict__ S2,
                                                 uint32_t* __restrict__ Skeep, int keepS, size_t npix, int D, int minD,
                                                 int uniq, int16_t* __restrict__ sel_d16, uint32_t* __restrict__ sel_key)
{
    const int lane = threadIdx.x & 63;
    const size_t pix = (size_t)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (pix >= npix) return;
    const size_t o = pix * (64 * NP) + lane * NP;
    const us2 cap = pk_splat(0x7FFF);
    us2 sv[NP];
#pragma unroll
    for (int q = 0; q < NP; ++q) sv[q] = pk_min(pk_adds(as_us2(S[o + q]), as_us2(S2[o + q])), cap);
    if (keepS) {
#pragma unroll
        for (int q = 0; q < NP; ++q) Skeep[o + q] = as_u32(sv[q]);
    }
    wta_select<NP>(sv, lane, D, minD, uniq, sel_d16 + pix, sel_key + pix);
}

template <int NP>
static int launch_trio_np(wass_ctx* c, const SgmDims& d, uint32_t* Sout, unsigned long long* halo, int xdir, int ydir, bool has_v,
                          hipStream_t stream)
{
    constexpr int W = 4;
    const int nstrips = (d.width1 + W - 1) / W;
    const uint32_t* C = (const uint32_t*)c->C.p;
    if (has_v)
        hipLaunchKernelGGL((k_trio<NP, W, true>), dim3(nstrips), dim3(64), 0, stream, C, Sout, halo, d.width1, d.h, xdir, ydir,
                           d.P1, d.P2, nstrips, (uint32_t*)c->flags.p);
    else
        hipLaunchKernelGGL((k_trio<NP, W, false>), dim3(nstrips), dim3(64), 0, stream, C, Sout, halo, d.width1, d.h, xdir, ydir,
                           d.P1, d.P2, nstrips, (uint32_t*)c->flags.p);
    WASS_HIP(c, hipGetLastError());
    return WASS_OK;
}

size_t trio_halo_bytes(const SgmDims& d)
{
    constexpr int W = 4;
    const size_t nstrips = (d.width1 + W - 1) / W;
    return nstrips * (size_t)d.h * 2 * ((d.NP + 1) / 2) * 64 * sizeof(unsigned long long);
}

int launch_trio(wass_ctx* c, const SgmDims& d, uint32_t* Sout, unsigned long long* halo, int xdir, int ydir, bool has_v,
                hipStream_t stream)
{
    switch (d.NP) {
        case 1: return launch_trio_np<1>(c, d, Sout, halo, xdir, ydir, has_v, stream);
        case 2: return launch_trio_np<2>(c, d, Sout, halo, xdir, ydir, has_v, stream);
        case 3: return launch_trio_np<3>(c, d, Sout, halo, xdir, ydir, has_v, stream);
        case 4: return launch_trio_np<4>(c, d, Sout, halo, xdir, ydir, has_v, stream);
        case 5: return launch_trio_np<5>(c, d, Sout, halo, xdir, ydir, has_v, stream);
        case 6: return launch_trio_np<6>(c, d, Sout, halo, xdir, ydir, has_v, stream);
        case 7: return launch_trio_np<7>(c, d, Sout, halo, xdir, ydir, has_v, stream);
        case 8: return launch_trio_np<8>(c, d, Sout, halo, xdir, ydir, has_v, stream);
    }
    return set_err(c, WASS_ERR_UNSUPPORTED, "MAX_DISPARITY %d not supported (max 1024)", d.D);
}

int launch_wta_sum(wass_ctx* c, const SgmDims& d, const uint32_t* S, const uint32_t* S2, hipStream_t stream)
{
    const size_t npix = (size_t)d.width1 * d.h;
    const dim3 grid((unsigned)((npix + 3) / 4)), blk(256);
    int16_t* sd = (int16_t*)c->sel_d16.p;
    uint32_t* sk = (uint32_t*)c->sel_key.p;
#define WASS_WS(NPV) hipLaunchKernelGGL(k_wta_sum<NPV>, grid, blk, 0, stream, S, S2, (uint32_t*)c->S.p, c->debug ? 1 : 0, npix, d.D, d.minD, d.uniq, sd, sk)
    switch (d.NP) {
        case 1: WASS_WS(1); break; case 2: WASS_WS(2); break; case 3: WASS_WS(3); break; case 4: WASS_WS(4); break;
        case 5: WASS_WS(5); break; case 6: WASS_WS(6); break; case 7: WASS_WS(7); break; case 8: WASS_WS(8); break;
        default: return set_err(c, WASS_ERR_UNSUPPORTED, "MAX_DISPARITY %d not supported (max 1024)", d.D);
    }
#undef WASS_WS
    WASS_HIP(c, hipGetLastError());
    return WASS_OK;
}

}  // namespace wass
