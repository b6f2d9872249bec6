// sgm_step.h -- the path-cost recurrence and its helpers, shared by the aggregation kernels (sgm_aggregate.hip) and the
// column walk of the cost stage (sgm_cost.hip).
#pragma once

#include "common.h"

namespace wass {

// State carried along one chain: the (un-normalised) path costs of the previous
// pixel and their minimum over d.  Keeping the minimum as a separate wave-uniform
// scalar takes its cross-lane reduction off the critical path of the next step:
//   L'(d) = C(d) + min(L(d), min(L(d-1), L(d+1)) + P1, m + P2) - m,   m' = min_d L'
// (same value as the normalised form in the header comment; only L - m matters).
// Streamed once per kernel, far larger than any cache: non-temporal hints measured +3..6 % on this access pattern
// (scripts/micro/nt.hip).  WASS_NT=0 at build time keeps the plain forms for comparison.
#ifndef WASS_NT
#define WASS_NT 1
#endif
__device__ __forceinline__ uint32_t ld_stream(const uint32_t* p)
{
#if WASS_NT
    return __builtin_nontemporal_load(p);
#else
    return *p;
#endif
}
__device__ __forceinline__ void st_stream(uint32_t* p, uint32_t v)
{
#if WASS_NT
    __builtin_nontemporal_store(v, p);
#else
    *p = v;
#endif
}
// NP consecutive dwords of one lane (address aligned to 4*NP bytes) as the widest vector accesses that alignment allows
typedef uint32_t wass_u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t wass_u32x4 __attribute__((ext_vector_type(4)));
template <int NP>
__device__ __forceinline__ void ld_stream_vec(const uint32_t* __restrict__ p, us2 (&dst)[NP])
{
    if constexpr (NP % 4 == 0) {
#pragma unroll
        for (int j = 0; j < NP; j += 4) {
#if WASS_NT
            const wass_u32x4 v = __builtin_nontemporal_load((const wass_u32x4*)(p + j));
#else
            const wass_u32x4 v = *(const wass_u32x4*)(p + j);
#endif
            dst[j] = as_us2(v.x); dst[j + 1] = as_us2(v.y); dst[j + 2] = as_us2(v.z); dst[j + 3] = as_us2(v.w);
        }
    } else if constexpr (NP % 2 == 0) {
#pragma unroll
        for (int j = 0; j < NP; j += 2) {
#if WASS_NT
            const wass_u32x2 v = __builtin_nontemporal_load((const wass_u32x2*)(p + j));
#else
            const wass_u32x2 v = *(const wass_u32x2*)(p + j);
#endif
            dst[j] = as_us2(v.x); dst[j + 1] = as_us2(v.y);
        }
    } else {
#pragma unroll
        for (int j = 0; j < NP; ++j) dst[j] = as_us2(ld_stream(p + j));
    }
}
template <int NP>
__device__ __forceinline__ void st_stream_vec(uint32_t* __restrict__ p, const us2 (&src)[NP])
{
    if constexpr (NP % 4 == 0) {
#pragma unroll
        for (int j = 0; j < NP; j += 4) {
            wass_u32x4 v = { as_u32(src[j]), as_u32(src[j + 1]), as_u32(src[j + 2]), as_u32(src[j + 3]) };
#if WASS_NT
            __builtin_nontemporal_store(v, (wass_u32x4*)(p + j));
#else
            *(wass_u32x4*)(p + j) = v;
#endif
        }
    } else if constexpr (NP % 2 == 0) {
#pragma unroll
        for (int j = 0; j < NP; j += 2) {
            wass_u32x2 v = { as_u32(src[j]), as_u32(src[j + 1]) };
#if WASS_NT
            __builtin_nontemporal_store(v, (wass_u32x2*)(p + j));
#else
            *(wass_u32x2*)(p + j) = v;
#endif
        }
    } else {
#pragma unroll
        for (int j = 0; j < NP; ++j) st_stream(p + j, as_u32(src[j]));
    }
}

// ---------------------------------------------------------------------------
// Buffer addressing for the chain kernels: address = descriptor base (SGPRs, re-based per segment with scalar
// arithmetic) + scalar byte offset + per-lane byte offset.  global_load/store with a runtime stride needs a 64-bit VALU
// add per access (v_lshl_add_u64: 5 % of the pair kernel's vector instructions); buffer_load ... offen takes the scalar
// offset as an operand and needs none.  Raw buffer (stride 0), 4 GiB window, no swizzle; aux 2 = nt.
// ---------------------------------------------------------------------------
typedef __amdgpu_buffer_rsrc_t rsrc_t;
typedef uint32_t wass_v2u __attribute__((__vector_size__(2 * sizeof(uint32_t))));
typedef uint32_t wass_v4u __attribute__((__vector_size__(4 * sizeof(uint32_t))));
__device__ __forceinline__ rsrc_t mk_rsrc(const void* p)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0xFFFFFFFF, 0x00020000);
}
template <int NP>
__device__ __forceinline__ void buf_ld(rsrc_t r, uint32_t voff, uint32_t soff, us2 (&dst)[NP])
{
    if constexpr (NP % 4 == 0) {
#pragma unroll
        for (int j = 0; j < NP; j += 4) {
            const wass_v4u v = __builtin_amdgcn_raw_buffer_load_b128(r, voff + 4 * j, soff, WASS_NT ? 2 : 0);
            dst[j] = as_us2(v[0]); dst[j + 1] = as_us2(v[1]); dst[j + 2] = as_us2(v[2]); dst[j + 3] = as_us2(v[3]);
        }
    } else if constexpr (NP % 2 == 0) {
#pragma unroll
        for (int j = 0; j < NP; j += 2) {
            const wass_v2u v = __builtin_amdgcn_raw_buffer_load_b64(r, voff + 4 * j, soff, WASS_NT ? 2 : 0);
            dst[j] = as_us2(v[0]); dst[j + 1] = as_us2(v[1]);
        }
    } else {
#pragma unroll
        for (int j = 0; j < NP; ++j) dst[j] = as_us2(__builtin_amdgcn_raw_buffer_load_b32(r, voff + 4 * j, soff, WASS_NT ? 2 : 0));
    }
}
template <int NP>
__device__ __forceinline__ void buf_st(rsrc_t r, uint32_t voff, uint32_t soff, const us2 (&src)[NP])
{
    // A buffer store of MORE than 64 bits whose scalar offset is an SGPR fetches its data registers late (the ISA manual
    // lists buffer_store_dwordx3/x4 with an SGPR offset as needing a wait state before a VALU overwrites the data; under
    // memory back-pressure the window is far longer and an LDS read or a load landing in the same registers also
    // corrupts the store -- measured: ~1e-4 of the pixels wrong at D = 512, run to run different, only with x4 stores,
    // never with x4 loads or x2 stores).  It does not take a 128-bit store in the SOURCE to get one: the compiler merges
    // neighbouring dword / dwordx2 stores (NP = 3: three b32 stores became buffer_store_dwordx3 ... s8 offen, and the third
    // dword of a handful of vectors per frame came out wrong in k_pairx<3>, round 4).  Whenever a vector is wider than
    // one 64-bit store the offset therefore goes into the VGPR: one v_add per vector.
#ifndef WASS_BUFST_SGPR_WIDE
#define WASS_BUFST_SGPR_WIDE 0                     // 1: the form that runs into the trap (kept selectable: scripts/selftest.py on such
#endif                                             // a build is the demonstration that the device self-test catches it)
    if constexpr (NP > 2 && WASS_BUFST_SGPR_WIDE) {
#pragma unroll
        for (int j = 0; j < NP; ++j) __builtin_amdgcn_raw_buffer_store_b32(as_u32(src[j]), r, voff + 4 * j, soff, WASS_NT ? 2 : 0);
    } else if constexpr (NP > 2) {
        const uint32_t vo = voff + soff;
        if constexpr (NP % 4 == 0) {
#pragma unroll
            for (int j = 0; j < NP; j += 4) {
                const wass_v4u v = { as_u32(src[j]), as_u32(src[j + 1]), as_u32(src[j + 2]), as_u32(src[j + 3]) };
                __builtin_amdgcn_raw_buffer_store_b128(v, r, vo + 4 * j, 0, WASS_NT ? 2 : 0);
            }
        } else if constexpr (NP % 2 == 0) {
#pragma unroll
            for (int j = 0; j < NP; j += 2) {
                const wass_v2u v = { as_u32(src[j]), as_u32(src[j + 1]) };
                __builtin_amdgcn_raw_buffer_store_b64(v, r, vo + 4 * j, 0, WASS_NT ? 2 : 0);
            }
        } else {
#pragma unroll
            for (int j = 0; j < NP; ++j) __builtin_amdgcn_raw_buffer_store_b32(as_u32(src[j]), r, vo + 4 * j, 0, WASS_NT ? 2 : 0);
        }
    } else if constexpr (NP == 2) {
        const wass_v2u v = { as_u32(src[0]), as_u32(src[1]) };
        __builtin_amdgcn_raw_buffer_store_b64(v, r, voff, soff, WASS_NT ? 2 : 0);
    } else {
        __builtin_amdgcn_raw_buffer_store_b32(as_u32(src[0]), r, voff, soff, WASS_NT ? 2 : 0);
    }
}

// One chain as the kernels address it: element t lies at pixel pix0 + t * pixstep.  A run of `count` elements starting
// at element `first` is addressed from its lowest address (descriptor base) with the scalar offset bias + e * sstep for
// its e-th element (sstep is negative for chains that walk towards lower addresses).
struct ChainAddr {
    long long pix0, pixstep;
    int sstep;                       // pixstep * bytes per vector
    template <int NP>
    __device__ __forceinline__ rsrc_t run(const uint32_t* vol, long long first, int count) const
    {
        const long long lp = pix0 + (first + (pixstep < 0 ? count - 1 : 0)) * pixstep;
        return mk_rsrc(vol + lp * (64 * NP));
    }
    __device__ __forceinline__ uint32_t bias(int count) const { return pixstep < 0 ? (uint32_t)(-(count - 1) * sstep) : 0u; }
};

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
        us2 n[NP];
#pragma unroll
        for (int j = 0; j < NP; ++j) n[j] = L[j] - mv;
        st_stream_vec<NP>(p, n);
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

// The K minima of a checkpointed segment (wave-uniform scalars) -> one record of K u16 in HBM, written as K/2 dwords
// by lanes 0 .. K/2-1 (a handful of instructions per SEGMENT; the pair kernel reads the record back with scalar loads).
template <int K>
__device__ __forceinline__ void store_minima(uint32_t* __restrict__ rec, const uint32_t (&ms)[K], int lane)
{
    static_assert(K % 2 == 0, "records are dwords");
    uint32_t v = ms[0] | (ms[1] << 16);
#pragma unroll
    for (int i = 1; i < K / 2; ++i) v = lane == i ? (ms[2 * i] | (ms[2 * i + 1] << 16)) : v;
    if (lane < K / 2) rec[lane] = v;
}

// The main-loop step of k_pair: chain a is a forward RE-computation whose minimum before the step is known (am: the
// checkpoint sweep recorded it, sgm_aggregate.hip / k_vsum_col), chain b is the backward path proper.  Same arithmetic as
// sgm_step_pair; chain a simply has no cross-lane reduction (6 DPP stages + v_readlane per step saved).
template <int NP>
__device__ __forceinline__ void sgm_step_fb(PathState<NP>& a, const uint32_t am, const us2 (&ca)[NP], us2 (&La)[NP],
                                            PathState<NP>& b, const us2 (&cb)[NP], us2 (&Lb)[NP],
                                            const us2 P1v, const uint32_t P2)
{
    a.shr = dpp_mov<DPP_WAVE_SHR1>(a.shr, as_u32(a.L[NP - 1]));
    b.shr = dpp_mov<DPP_WAVE_SHR1>(b.shr, as_u32(b.L[NP - 1]));
    a.shl = dpp_mov<DPP_WAVE_SHL1>(a.shl, as_u32(a.L[0]));
    b.shl = dpp_mov<DPP_WAVE_SHL1>(b.shl, as_u32(b.L[0]));
    const us2 amv = pk_splat(am), amp2 = pk_splat(am + P2);
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
    us2 bm = Lb[0];
#pragma unroll
    for (int j = 1; j < NP; ++j) bm = pk_min(bm, Lb[j]);
    b.m = wave_min_u32(min((uint32_t)bm.x, (uint32_t)bm.y));
#pragma unroll
    for (int j = 0; j < NP; ++j) { a.L[j] = La[j]; b.L[j] = Lb[j]; }
}

// Two forward RE-computations side by side (the row paths inside a block of k_pairx): both minima are known scalars,
// no reduction at all.
template <int NP>
__device__ __forceinline__ void sgm_step_ff(PathState<NP>& a, const uint32_t am, const us2 (&ca)[NP], us2 (&La)[NP],
                                            PathState<NP>& b, const uint32_t bm, const us2 (&cb)[NP], us2 (&Lb)[NP],
                                            const us2 P1v, const uint32_t P2)
{
    a.shr = dpp_mov<DPP_WAVE_SHR1>(a.shr, as_u32(a.L[NP - 1]));
    b.shr = dpp_mov<DPP_WAVE_SHR1>(b.shr, as_u32(b.L[NP - 1]));
    a.shl = dpp_mov<DPP_WAVE_SHL1>(a.shl, as_u32(a.L[0]));
    b.shl = dpp_mov<DPP_WAVE_SHL1>(b.shl, as_u32(b.L[0]));
    const us2 amv = pk_splat(am), amp2 = pk_splat(am + P2);
    const us2 bmv = pk_splat(bm), bmp2 = pk_splat(bm + P2);
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
#pragma unroll
    for (int j = 0; j < NP; ++j) { a.L[j] = La[j]; b.L[j] = Lb[j]; }
}
// ... and one alone (partial blocks)
template <int NP>
__device__ __forceinline__ void sgm_step_f(PathState<NP>& a, const uint32_t am, const us2 (&ca)[NP], us2 (&La)[NP], const us2 P1v,
                                           const uint32_t P2)
{
    a.shr = dpp_mov<DPP_WAVE_SHR1>(a.shr, as_u32(a.L[NP - 1]));
    a.shl = dpp_mov<DPP_WAVE_SHL1>(a.shl, as_u32(a.L[0]));
    const us2 amv = pk_splat(am), amp2 = pk_splat(am + P2);
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        const uint32_t alo = j == 0 ? a.shr : as_u32(a.L[j - 1]);
        const uint32_t ahi = j == NP - 1 ? a.shl : as_u32(a.L[j + 1]);
        const us2 anl = as_us2(__builtin_amdgcn_alignbit(as_u32(a.L[j]), alo, 16));
        const us2 anr = as_us2(__builtin_amdgcn_alignbit(ahi, as_u32(a.L[j]), 16));
        us2 ax = pk_adds(pk_min(anl, anr), P1v);
        ax = pk_min(pk_min(a.L[j], ax), amp2) - amv;
        La[j] = pk_adds(ca[j], ax);
    }
#pragma unroll
    for (int j = 0; j < NP; ++j) a.L[j] = La[j];
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

// Meet-in-the-middle split of a chain family (rows / columns have only ~2 chains per SIMD): sub-chain c2 = 2c + half is
// the first n/2 pixels of chain c walked in the family's direction (half 0), or the remaining ones walked from the far end
// in the opposite direction (half 1).  Both start at an image border with the all-zero state, so their checkpoint sweeps
// are independent; the state each one ends with is exactly the state the OTHER half's backward sweep starts from, so
// the two pair kernels are independent too: twice the waves for the same bytes.
__device__ __forceinline__ void half_chain_geometry(int c2, int& dx, int& dy, int width1, int h, int& x0, int& y0, int& n)
{
    chain_geometry(c2 >> 1, dx, dy, width1, h, x0, y0, n);
    const int mid = n / 2;
    if (c2 & 1) { x0 += (n - 1) * dx; y0 += (n - 1) * dy; dx = -dx; dy = -dy; n -= mid; }
    else n = mid;
}

// K consecutive vectors of a chain -> registers; GUARD: only the first len exist
template <int NP, int K, bool GUARD>
__device__ __forceinline__ void load_seg(const uint32_t* __restrict__ p, long long step, int len, us2 (&dst)[K][NP])
{
#pragma unroll
    for (int u = 0; u < K; ++u)
        if (!GUARD || u < len) {
            ld_stream_vec<NP>(p + u * step, dst[u]);
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
// S at a wave-uniform disparity d: the whole lane vector is fetched with NP v_readlane and the half-word is
// picked on the scalar unit.
template <int NP>
__device__ __forceinline__ int s_at(const us2 (&Sv)[NP], int d)
{
    const int ln = d / (2 * NP), slot = d % (2 * NP);       // wave-uniform
    uint32_t pv = 0;
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        const uint32_t w = (uint32_t)__builtin_amdgcn_readlane((int)as_u32(Sv[j]), ln);
        pv = (slot >> 1) == j ? w : pv;
    }
    return (int)((slot & 1) ? (pv >> 16) : (pv & 0xFFFF));
}

// trunc(num / den) for |num| < 2^24, 0 < den < 2^24 without the 35-instruction integer-division expansion:
// float reciprocal estimate, then one correction step each way (the estimate is off by at most one).
__device__ __forceinline__ int div_small(int num, int den)
{
    const int an = num < 0 ? -num : num;
    int q = (int)((float)an * __builtin_amdgcn_rcpf((float)den));
    const int r = an - q * den;
    q += (r >= den) - (r < 0);
    return num < 0 ? -q : q;
}

// Winner-take-all for the K finished S vectors of one chain segment, written stage by stage over the K cells so
// that the K independent reductions / readlanes / scalar tails overlap (one cell at a time the instruction stream
// is a single dependent chain full of DPP and readlane wait states).  Branch-free; lane u collects the result of
// cell u (res_d: fixed-point disparity, res_k: (minS << 16) | d or ~0 for a rejected pixel), so that a batch is
// written with one store per output array.
#ifndef WASS_WTA_VECTOR_TAIL
#define WASS_WTA_VECTOR_TAIL 1
#endif
#if WASS_WTA_VECTOR_TAIL
// Round 4: the per-cell tail -- neighbours of the winner, uniqueness decision, sub-pixel division, output words -- used to
// run cell by cell on the scalar unit (88 scalar + 34 vector instructions per cell; the last aggregation kernel carries 99
// scalar instructions per pixel, which is a CU's whole scalar issue for 0.8 of its 1.2 ms).  Now only what NEEDS the
// scalar unit stays there (the reduced key, the ballots of the uniqueness count, the lane index of the two neighbours); the
// per-cell values are gathered into lane u and the tail runs ONCE per batch, K cells side by side in K lanes.  Same integer
// arithmetic, same results.
template <int NP, int K>
__device__ __forceinline__ void wta_batch_eval(const us2 (&Sv)[K][NP], int lane, int D, int minD, int uniq, int& res_d,
                                               uint32_t& res_k)
{
    constexpr int V = 2 * NP;
    constexpr int ABSENT = 1 << 22;                             // a neighbour that does not exist never passes a test
    const int dlane = lane * V;
    // padded slots (d >= D) always hold the saturated 0x7FFF and a larger d than every real slot, so they can
    // only win the key minimum when every real slot is 0x7FFF too -- and then the smallest (real) d wins anyway
    uint32_t key[K];
#pragma unroll
    for (int u = 0; u < K; ++u) {
        key[u] = 0xFFFFFFFFu;
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            const uint32_t w = as_u32(Sv[u][j]);
            key[u] = min(key[u], min((w << 16) | (uint32_t)(dlane + 2 * j), (w & 0xFFFF0000u) | (uint32_t)(dlane + 2 * j + 1)));
        }
    }
#pragma unroll
    for (int u = 0; u < K; ++u) key[u] = wave_min_u32(key[u]);
    const int q = 100 - uniq;
    // lanes whose slot j is a real disparity (d = lane*V + j < D), as ballot masks
    unsigned long long inr[V];
#pragma unroll
    for (int j = 0; j < V; ++j) {
        const int nl = (D - j + V - 1) / V;
        inr[j] = nl >= 64 ? ~0ull : ((1ull << (nl < 0 ? 0 : nl)) - 1ull);
    }
    // cell u's values, gathered into lane u
    uint32_t g_key = 0xFFFFFFFFu, g_aw[NP], g_cw[NP];
    int g_cnt = 0;
#pragma unroll
    for (int j = 0; j < NP; ++j) g_aw[j] = g_cw[j] = 0;
#pragma unroll
    for (int u = 0; u < K; ++u) {
        const int minS = (int)(key[u] >> 16);
        const int best = minS >= 32767 ? -1 : (int)(key[u] & 0xFFFF);
        // uniqueness: the slots with S[d]*(100-uniq) < minS*100 over the whole vector (ballots, scalar popcounts)
        const int T = minS * 100;
        int cnt = 0;
#pragma unroll
        for (int j = 0; j < V; ++j) {
            const uint32_t sv = (j & 1) ? Sv[u][j >> 1].y : Sv[u][j >> 1].x;
            cnt += __builtin_popcountll(__builtin_amdgcn_ballot_w64((int)__umul24(sv, (uint32_t)q) < T) & inr[j]);
        }
        const int lna = max(best - 1, 0) / V, lnc = min(best + 1, D - 1) / V;        // wave-uniform lane indices
        const bool mine = lane == u;
        g_key = mine ? key[u] : g_key;
        g_cnt = mine ? cnt : g_cnt;
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            const uint32_t aw = (uint32_t)__builtin_amdgcn_readlane((int)as_u32(Sv[u][j]), lna);
            const uint32_t cw = (uint32_t)__builtin_amdgcn_readlane((int)as_u32(Sv[u][j]), lnc);
            g_aw[j] = mine ? aw : g_aw[j];
            g_cw[j] = mine ? cw : g_cw[j];
        }
    }
    // ---- the tail, once, cell u in lane u
    const int minS = (int)(g_key >> 16);
    // "if (Sval < minS)" with minS initialised to MAX_COST never fires when every S is MAX_COST
    const int best = minS >= 32767 ? -1 : (int)(g_key & 0xFFFF);
    const int T = minS * 100;
    auto half_at = [&](const uint32_t (&w)[NP], int d) {
        const int slot = d % V;
        uint32_t pv = 0;
#pragma unroll
        for (int j = 0; j < NP; ++j) pv = (slot >> 1) == j ? w[j] : pv;
        return (int)((slot & 1) ? (pv >> 16) : (pv & 0xFFFF));
    };
    const int am = half_at(g_aw, max(best - 1, 0)), cp = half_at(g_cw, min(best + 1, D - 1));
    const int a = best >= 1 ? am : ABSENT, cc = best + 1 < D ? cp : ABSENT;
    const int near = (a * q < T) + (cc * q < T) + (best >= 0 && minS * q < T);
    const bool ok = g_cnt <= near, sub = 0 < best && best < D - 1;
    const int denom2 = max(a + cc - 2 * minS, 1);
    const int frac = div_small((a - cc) * 16 + denom2, denom2 * 2);
    res_d = ok ? (best * 16 + (sub ? frac : 0) + minD * 16) : (minD - 1) * 16;
    res_k = ok ? (((uint32_t)minS << 16) | (uint32_t)(best & 0xFFFF)) : 0xFFFFFFFFu;
}
#else
template <int NP, int K>
__device__ __forceinline__ void wta_batch_eval(const us2 (&Sv)[K][NP], int lane, int D, int minD, int uniq, int& res_d,
                                               uint32_t& res_k)
{
    constexpr int V = 2 * NP;
    constexpr int ABSENT = 1 << 22;                             // a neighbour that does not exist never passes a test
    const int dlane = lane * V;
    // padded slots (d >= D) always hold the saturated 0x7FFF and a larger d than every real slot, so they can
    // only win the key minimum when every real slot is 0x7FFF too -- and then the smallest (real) d wins anyway
    uint32_t key[K];
#pragma unroll
    for (int u = 0; u < K; ++u) {
        key[u] = 0xFFFFFFFFu;
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            const uint32_t w = as_u32(Sv[u][j]);
            key[u] = min(key[u], min((w << 16) | (uint32_t)(dlane + 2 * j), (w & 0xFFFF0000u) | (uint32_t)(dlane + 2 * j + 1)));
        }
    }
#pragma unroll
    for (int u = 0; u < K; ++u) key[u] = wave_min_u32(key[u]);
    const int q = 100 - uniq;
    // lanes whose slot j is a real disparity (d = lane*V + j < D), as ballot masks
    unsigned long long inr[V];
#pragma unroll
    for (int j = 0; j < V; ++j) {
        const int nl = (D - j + V - 1) / V;
        inr[j] = nl >= 64 ? ~0ull : ((1ull << (nl < 0 ? 0 : nl)) - 1ull);
    }
    res_d = 0;
    res_k = 0;
    // the scalar tails run cell by cell (few live SGPRs; batching them spills scalars into VGPR lanes)
#pragma unroll
    for (int u = 0; u < K; ++u) {
        const int minS = (int)(key[u] >> 16);
        // "if (Sval < minS)" with minS initialised to MAX_COST never fires when every S is MAX_COST
        const int best = minS >= 32767 ? -1 : (int)(key[u] & 0xFFFF);
        // uniqueness: some d outside best-1..best+1 with S[d]*(100-uniq) < minS*100.  Count the slots that satisfy
        // the inequality over the whole vector (ballots, scalar popcounts) and subtract the up to three
        // neighbours, which are wave-uniform values.
        const int T = minS * 100;
        int cnt = 0;
#pragma unroll
        for (int j = 0; j < V; ++j) {
            const uint32_t sv = (j & 1) ? Sv[u][j >> 1].y : Sv[u][j >> 1].x;
            cnt += __builtin_popcountll(__builtin_amdgcn_ballot_w64((int)__umul24(sv, (uint32_t)q) < T) & inr[j]);
        }
        const int am = s_at<NP>(Sv[u], max(best - 1, 0)), cp = s_at<NP>(Sv[u], min(best + 1, D - 1));
        const int a = best >= 1 ? am : ABSENT, cc = best + 1 < D ? cp : ABSENT;
        const int near = (a * q < T) + (cc * q < T) + (best >= 0 && minS * q < T);
        const bool ok = cnt <= near, sub = 0 < best && best < D - 1;
        const int denom2 = max(a + cc - 2 * minS, 1);
        const int frac = div_small((a - cc) * 16 + denom2, denom2 * 2);
        const int out = ok ? (best * 16 + (sub ? frac : 0) + minD * 16) : (minD - 1) * 16;
        const uint32_t k = ok ? (((uint32_t)minS << 16) | (uint32_t)(best & 0xFFFF)) : 0xFFFFFFFFu;
        res_d = lane == u ? out : res_d;
        res_k = lane == u ? k : res_k;
    }
}

#endif

// ... for cells that lie on an arithmetic progression of pixels (one chain segment)
template <int NP, int K>
__device__ __forceinline__ void wta_batch(const us2 (&Sv)[K][NP], int nvalid, int lane, int D, int minD, int uniq,
                                          int16_t* __restrict__ out_d16, uint32_t* __restrict__ out_key, long long pix0,
                                          long long pixstep)
{
    int res_d;
    uint32_t res_k;
    wta_batch_eval<NP, K>(Sv, lane, D, minD, uniq, res_d, res_k);
    if (lane < nvalid) {
        const long long px = pix0 + lane * pixstep;
        out_d16[px] = (int16_t)res_d;
        out_key[px] = res_k;
    }
}

// single-cell form
template <int NP>
__device__ __forceinline__ void wta_select(const us2 (&Sv)[NP], int lane, int D, int minD, int uniq,
                                           int16_t* __restrict__ out_d16, uint32_t* __restrict__ out_key)
{
    us2 one[1][NP];
#pragma unroll
    for (int j = 0; j < NP; ++j) one[0][j] = Sv[j];
    wta_batch<NP, 1>(one, 1, lane, D, minD, uniq, out_d16, out_key, 0, 0);
}


}  // namespace wass
