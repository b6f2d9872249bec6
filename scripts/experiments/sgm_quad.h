// sgm_quad.h -- the path recurrence in "quad" layout: FOUR chains per wavefront.
//
// Measured on MI355X (scripts/micro/valu2.hip): every VALU instruction of the recurrence -- v_pk_*_u16, v_min_u32,
// DPP moves alike -- issues at ~4.5 cycles per wave, so the kernels are bound by their instruction COUNT.  In the
// one-chain-per-wave layout of sgm_step.h a third of the instructions of a step are the cross-lane minimum over 64
// lanes (6 DPP stages + v_readlane) and the two wave shifts, paid once per pixel.  Here a pixel's disparity vector
// lives on the 16 lanes of one DPP row (lane s holds the 8*NP consecutive disparities 8*NP*s ..., as NQ = 4*NP
// packed u16 pairs), and the four rows of a wave advance four independent chains in lock-step: the row minimum is
// 4 DPP stages (quad_perm, quad_perm, row_half_mirror, row_mirror) shared by four pixels, it lands in every lane of
// the row (no v_readlane, no SGPR round trip), and the d-1 / d+1 neighbours cross lanes with row_shr:1 / row_shl:1.
// Per pixel and step: ~21 instructions instead of ~32 at NP = 2.
//
// The HBM layout of the volumes is unchanged: lane s of a row reads/writes the 16*NP bytes at offset 16*NP*s of the
// pixel's vector.
#pragma once

#include "sgm_step.h"

namespace wass {

enum : int { DPP_ROW_SHL1 = 0x101, DPP_ROW_SHR1 = 0x111 };

// minimum over the 16 lanes of each DPP row, left in every lane of the row
__device__ __forceinline__ uint32_t row_min_u32(uint32_t v)
{
    v = min(v, dpp_mov<DPP_QUAD_XOR1>(0xFFFFFFFFu, v));
    v = min(v, dpp_mov<DPP_QUAD_XOR2>(0xFFFFFFFFu, v));
    v = min(v, dpp_mov<DPP_ROW_HALF_MIRROR>(0xFFFFFFFFu, v));
    v = min(v, dpp_mov<DPP_ROW_MIRROR>(0xFFFFFFFFu, v));
    return v;
}
__device__ __forceinline__ void row_min2_u32(uint32_t& a, uint32_t& b)
{
    a = min(a, dpp_mov<DPP_QUAD_XOR1>(0xFFFFFFFFu, a));
    b = min(b, dpp_mov<DPP_QUAD_XOR1>(0xFFFFFFFFu, b));
    a = min(a, dpp_mov<DPP_QUAD_XOR2>(0xFFFFFFFFu, a));
    b = min(b, dpp_mov<DPP_QUAD_XOR2>(0xFFFFFFFFu, b));
    a = min(a, dpp_mov<DPP_ROW_HALF_MIRROR>(0xFFFFFFFFu, a));
    b = min(b, dpp_mov<DPP_ROW_HALF_MIRROR>(0xFFFFFFFFu, b));
    a = min(a, dpp_mov<DPP_ROW_MIRROR>(0xFFFFFFFFu, a));
    b = min(b, dpp_mov<DPP_ROW_MIRROR>(0xFFFFFFFFu, b));
}

// minimum of N packed registers as a balanced tree (dependency depth log2 N instead of N - 1)
template <int N>
__device__ __forceinline__ us2 pk_min_tree(const us2 (&v)[N])
{
    us2 t[N];
#pragma unroll
    for (int j = 0; j < N; ++j) t[j] = v[j];
#pragma unroll
    for (int w = 1; w < N; w *= 2)
#pragma unroll
        for (int j = 0; j + w < N; j += 2 * w) t[j] = pk_min(t[j], t[j + w]);
    return t[0];
}

// per-lane 16-bit value (< 65536) -> both halves of a packed pair
__device__ __forceinline__ us2 pk_splat_v(uint32_t m) { return as_us2(m | (m << 16)); }

template <int NQ>
struct QState {
    us2 L[NQ];
    // Minimum of this lane's values of L, NOT yet reduced over the row: the four DPP stages of the row reduction are
    // issued by the NEXT step, spread between its first instructions, which do not need the minimum yet -- so the
    // reduction costs no wait states and sits on nobody's critical path.
    uint32_t pm;
    // destinations of the two row shifts: lane 0 (resp. 15) of every row has no source lane and keeps the 0xFFFF
    // sentinels of d = -1 / d = Dp
    uint32_t shr = 0xFFFFFFFFu, shl = 0xFFFFFFFFu;
    __device__ __forceinline__ void reset()
    {
#pragma unroll
        for (int j = 0; j < NQ; ++j) L[j] = pk_splat(0);
        pm = 0;
    }
    // costs relative to the row minimum (what a checkpoint / edge state stores)
    __device__ __forceinline__ void normalised(us2 (&n)[NQ]) const
    {
        const us2 mv = pk_splat_v(row_min_u32(pm));
#pragma unroll
        for (int j = 0; j < NQ; ++j) n[j] = L[j] - mv;
    }
    __device__ __forceinline__ void load_normalised(const us2 (&v)[NQ])
    {
#pragma unroll
        for (int j = 0; j < NQ; ++j) L[j] = v[j];
        pm = 0;                                  // the row minimum of a normalised state is 0
    }
};

// One step of the recurrence for the four chains of a wave, written stage by stage over the NQ registers: consecutive
// instructions are independent (the packed-math units need a wait state between dependent instructions, which the
// compiler otherwise fills with s_nop), and the row reduction of the previous step's minimum is threaded through the
// stages that do not depend on it.
template <int NQ>
__device__ __forceinline__ void qstep(QState<NQ>& st, const us2 (&c)[NQ], us2 (&Lo)[NQ], const us2 P1v, const us2 P2v)
{
    static_assert(NQ >= 4, "quad layout needs at least four registers per lane");
    st.shr = dpp_mov<DPP_ROW_SHR1>(st.shr, as_u32(st.L[NQ - 1]));
    st.shl = dpp_mov<DPP_ROW_SHL1>(st.shl, as_u32(st.L[0]));
    uint32_t m = st.pm;
    us2 nl[NQ], t[NQ];
#pragma unroll
    for (int j = 1; j < NQ; ++j) nl[j] = as_us2(__builtin_amdgcn_alignbit(as_u32(st.L[j]), as_u32(st.L[j - 1]), 16));   // (d-1, d)
    m = min(m, dpp_mov<DPP_QUAD_XOR1>(0xFFFFFFFFu, m));
    nl[0] = as_us2(__builtin_amdgcn_alignbit(as_u32(st.L[0]), st.shr, 16));
#pragma unroll
    for (int j = 0; j < NQ - 1; ++j) t[j] = as_us2(__builtin_amdgcn_alignbit(as_u32(st.L[j + 1]), as_u32(st.L[j]), 16));   // (d+1, d+2)
    m = min(m, dpp_mov<DPP_QUAD_XOR2>(0xFFFFFFFFu, m));
    t[NQ - 1] = as_us2(__builtin_amdgcn_alignbit(st.shl, as_u32(st.L[NQ - 1]), 16));
#pragma unroll
    for (int j = 0; j < NQ; ++j) t[j] = pk_min(t[j], nl[j]);
    m = min(m, dpp_mov<DPP_ROW_HALF_MIRROR>(0xFFFFFFFFu, m));
#pragma unroll
    for (int j = 0; j < NQ; ++j) t[j] = pk_adds(t[j], P1v);
    m = min(m, dpp_mov<DPP_ROW_MIRROR>(0xFFFFFFFFu, m));
#pragma unroll
    for (int j = 0; j < NQ; ++j) t[j] = pk_min(t[j], st.L[j]);
    const us2 mv = pk_splat_v(m), mp2 = mv + P2v;
#pragma unroll
    for (int j = 0; j < NQ; ++j) t[j] = pk_min(t[j], mp2);
#pragma unroll
    for (int j = 0; j < NQ; ++j) t[j] = t[j] - mv;
#pragma unroll
    for (int j = 0; j < NQ; ++j) Lo[j] = pk_adds(c[j], t[j]);
    const us2 mm = pk_min_tree<NQ>(Lo);
    st.pm = min((uint32_t)mm.x, (uint32_t)mm.y);
#pragma unroll
    for (int j = 0; j < NQ; ++j) st.L[j] = Lo[j];
}

// NQ consecutive dwords of one lane <-> registers (global memory, 16-byte aligned, streamed once)
template <int NQ>
__device__ __forceinline__ void q_ld(const uint32_t* __restrict__ p, us2 (&dst)[NQ])
{
#pragma unroll
    for (int j = 0; j < NQ; j += 4) {
        const wass_u32x4 v = __builtin_nontemporal_load((const wass_u32x4*)(p + j));
        dst[j] = as_us2(v.x); dst[j + 1] = as_us2(v.y); dst[j + 2] = as_us2(v.z); dst[j + 3] = as_us2(v.w);
    }
}
template <int NQ>
__device__ __forceinline__ void q_st(uint32_t* __restrict__ p, const us2 (&src)[NQ])
{
#pragma unroll
    for (int j = 0; j < NQ; j += 4) {
        wass_u32x4 v = { as_u32(src[j]), as_u32(src[j + 1]), as_u32(src[j + 2]), as_u32(src[j + 3]) };
        __builtin_nontemporal_store(v, (wass_u32x4*)(p + j));
    }
}

}  // namespace wass
