// micro-benchmark: issue rate of the instruction kinds the path recurrence is made of (v_pk_*_u16, v_alignbit, DPP moves,
// v_min_u32 with DPP, v_readlane) per SIMD, as a function of waves per SIMD.  Prints cycles per wave-instruction per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef unsigned short us2 __attribute__((ext_vector_type(2)));
#define N_IT 2000

template <int KIND>
__global__ void __launch_bounds__(256) k(uint32_t* out, int iters)
{
    uint32_t a[8];
    for (int i = 0; i < 8; ++i) a[i] = threadIdx.x * 2654435761u + i;
    const uint32_t c = blockIdx.x | 0x10001;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < 8; ++i) {       // 8 independent dependency chains
                if (KIND == 0) a[i] = __builtin_bit_cast(uint32_t, __builtin_elementwise_add_sat(__builtin_bit_cast(us2, a[i]), __builtin_bit_cast(us2, c)));
                if (KIND == 1) a[i] = __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(us2, a[i]), __builtin_bit_cast(us2, c + r)));
                if (KIND == 2) a[i] = __builtin_amdgcn_alignbit(a[i], a[(i + 1) & 7], 16);
                if (KIND == 3) a[i] = (uint32_t)__builtin_amdgcn_update_dpp((int)a[i], (int)a[(i + 1) & 7], 0x138, 0xf, 0xf, false);   // wave_shr:1
                if (KIND == 4) a[i] = min(a[i], (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)a[i], 0xB1, 0xf, 0xf, false));          // quad_perm + min
                if (KIND == 5) a[i] = a[i] + c;                                                                                            // plain v_add_u32
                if (KIND == 6) a[i] = a[i] * 3 + (uint32_t)__builtin_amdgcn_readlane((int)a[(i + 1) & 7], 63);                          // readlane + mad
                if (KIND == 7) a[i] = __builtin_bit_cast(uint32_t, __builtin_bit_cast(us2, a[i]) - __builtin_bit_cast(us2, c));          // v_pk_sub_u16
            }
    }
    uint32_t s = 0;
    for (int i = 0; i < 8; ++i) s ^= a[i];
    if (s == 0x12345678) out[0] = s;
}

template <int KIND>
void run(const char* name, uint32_t* out)
{
    printf("%-28s", name);
    for (int wps : { 1, 2, 4, 8 }) {
        // 256 CUs x 4 SIMDs; a 256-thread block puts one wave on each SIMD of a CU
        const int blocks = 256 * wps;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, 10);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, N_IT);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double instr_per_simd = (double)N_IT * 32 * wps;           // wave-instructions issued on one SIMD
        printf("  %d w/SIMD: %5.2f cyc/instr", wps, ms * 1e-3 * 2.4e9 / instr_per_simd);
    }
    printf("\n");
}

int main()
{
    uint32_t* out; hipMalloc(&out, 64);
    run<5>("v_add_u32", out);
    run<0>("v_pk_add_u16 (sat)", out);
    run<7>("v_pk_sub_u16", out);
    run<1>("v_pk_min_u16", out);
    run<2>("v_alignbit_b32", out);
    run<3>("v_mov_b32 dpp wave_shr", out);
    run<4>("v_min_u32 + dpp quad_perm", out);
    run<6>("v_readlane + v_mad", out);
    return 0;
}
