// micro-benchmark 2: issue rate of candidate 32-bit replacements for the packed-u16 recurrence (inline asm so the
// compiler cannot fuse or rewrite them).  cycles per wave-instruction per SIMD at 1/2/4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define N_IT 2000
#define OP8(STR)                                                              \
    _Pragma("unroll") for (int r = 0; r < 4; ++r) {                           \
        asm volatile(STR : "+v"(a[0]) : "v"(a[1]), "v"(c));                   \
        asm volatile(STR : "+v"(a[1]) : "v"(a[2]), "v"(c));                   \
        asm volatile(STR : "+v"(a[2]) : "v"(a[3]), "v"(c));                   \
        asm volatile(STR : "+v"(a[3]) : "v"(a[4]), "v"(c));                   \
        asm volatile(STR : "+v"(a[4]) : "v"(a[5]), "v"(c));                   \
        asm volatile(STR : "+v"(a[5]) : "v"(a[6]), "v"(c));                   \
        asm volatile(STR : "+v"(a[6]) : "v"(a[7]), "v"(c));                   \
        asm volatile(STR : "+v"(a[7]) : "v"(a[0]), "v"(c));                   \
    }

template <int KIND>
__global__ void __launch_bounds__(256) k(uint32_t* out, int iters)
{
    uint32_t a[8];
    for (int i = 0; i < 8; ++i) a[i] = threadIdx.x * 2654435761u + i;
    uint32_t c = blockIdx.x | 0x10001;
    asm volatile("" : "+v"(c));
    for (int it = 0; it < iters; ++it) {
        if (KIND == 0) OP8("v_min_u32 %0, %0, %1")
        if (KIND == 1) OP8("v_min3_u32 %0, %0, %1, %2")
        if (KIND == 2) OP8("v_add3_u32 %0, %0, %1, %2")
        if (KIND == 3) OP8("v_and_b32 %0, %1, %2")
        if (KIND == 4) OP8("v_lshl_or_b32 %0, %1, 16, %2")
        if (KIND == 5) OP8("v_perm_b32 %0, %0, %1, %2")
        if (KIND == 6) OP8("v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1")
        if (KIND == 7) OP8("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf")
        if (KIND == 8) OP8("v_min_u32_dpp %0, %1, %0 row_mirror row_mask:0xf bank_mask:0xf")
        if (KIND == 9) OP8("v_min_u32_dpp %0, %1, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
        if (KIND == 10) OP8("v_mad_u32_u16 %0, %1, 1, %0 op_sel:[1,0,0,0]")
        if (KIND == 11) OP8("v_min_u16 %0, %0, %1")
        if (KIND == 12) OP8("v_pk_min_u16 %0, %0, %1")
        if (KIND == 13) OP8("v_sub_u32 %0, %0, %1")
        if (KIND == 14) OP8("v_max_i32 %0, %0, %1")
        if (KIND == 15) OP8("v_add_u16 %0, %0, %1")
        if (KIND == 16) OP8("v_pk_add_u16 %0, %0, %1")
        if (KIND == 17) OP8("v_bfe_u32 %0, %1, 16, 16")
        if (KIND == 18) OP8("v_min_i16 %0, %0, %1")
        if (KIND == 19) OP8("v_min3_u16 %0, %0, %1, %2")
        if (KIND == 20) OP8("v_pk_mad_u16 %0, %0, %1, %2")
        if (KIND == 21) OP8("v_cndmask_b32 %0, %0, %1, vcc")
    }
    uint32_t s = 0;
    for (int i = 0; i < 8; ++i) s ^= a[i];
    if (s == 0x12345678) out[0] = s;
}

template <int KIND>
void run(const char* name, uint32_t* out)
{
    printf("%-34s", name);
    for (int wps : { 1, 2, 4 }) {
        const int blocks = 256 * wps;
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, 10);
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, N_IT);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        const double instr_per_simd = (double)N_IT * 32 * wps;
        printf("  %d w/SIMD: %5.2f", wps, ms * 1e-3 * 2.4e9 / instr_per_simd);
    }
    printf("   (cyc/instr @2.4GHz)\n");
}

int main()
{
    uint32_t* out; (void)hipMalloc(&out, 64);
    run<0>("v_min_u32", out); run<13>("v_sub_u32", out); run<14>("v_max_i32", out);
    run<1>("v_min3_u32", out); run<2>("v_add3_u32", out);
    run<3>("v_and_b32", out); run<4>("v_lshl_or_b32", out); run<5>("v_perm_b32", out); run<17>("v_bfe_u32", out);
    run<6>("v_add_u32_sdwa src1:WORD_1", out); run<10>("v_mad_u32_u16 op_sel", out);
    run<11>("v_min_u16", out); run<15>("v_add_u16", out); run<18>("v_min_i16", out); run<19>("v_min3_u16", out);
    run<12>("v_pk_min_u16", out); run<16>("v_pk_add_u16", out); run<20>("v_pk_mad_u16", out);
    run<7>("v_mov_b32_dpp row_shr:1", out); run<8>("v_min_u32_dpp row_mirror", out); run<9>("v_min_u32_dpp quad_perm", out);
    run<21>("v_cndmask_b32", out);
    return 0;
}
