// micro-benchmark: do non-temporal loads / stores change the achievable bandwidth of the chain access pattern?
// (one wave per column, 512-byte vectors, stride = one image row; whole 2.5 GB volume, beyond every cache)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int MODE, bool NTL, bool NTS>   // 0 read, 1 read A + write B, 2 read-modify-write A
__global__ void __launch_bounds__(256) k_cols(uint2* __restrict__ A, uint2* __restrict__ B, long long W, long long R, unsigned* sink)
{
    const int lane = threadIdx.x & 63;
    const long long c = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (c >= W) return;
    uint2* a = A + c * 64 + lane;
    uint2* b = B + c * 64 + lane;
    unsigned acc = 0;
    for (long long k0 = 0; k0 + 8 <= R; k0 += 8) {
        uint2 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const uint2* p = a + (k0 + u) * W * 64;
            if (NTL) { v[u].x = __builtin_nontemporal_load(&p->x); v[u].y = __builtin_nontemporal_load(&p->y); }
            else v[u] = *p;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (MODE == 0) { acc += v[u].x; continue; }
            uint2 o = v[u]; o.x += 1;
            uint2* q = (MODE == 1 ? b : a) + (k0 + u) * W * 64;
            if (NTS) { __builtin_nontemporal_store(o.x, &q->x); __builtin_nontemporal_store(o.y, &q->y); }
            else *q = o;
        }
    }
    if (MODE == 0 && acc == 0xdeadbeef) *sink = acc;
}

template <int MODE, bool NTL, bool NTS>
static double run(void* A, void* B, long long W, long long R, unsigned* sink)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const dim3 grid((unsigned)((W + 3) / 4)), block(256);
    hipLaunchKernelGGL((k_cols<MODE, NTL, NTS>), grid, block, 0, 0, (uint2*)A, (uint2*)B, W, R, sink);
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((k_cols<MODE, NTL, NTS>), grid, block, 0, 0, (uint2*)A, (uint2*)B, W, R, sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double gb = (double)R * W * 512 / 1e9 * (MODE == 0 ? 1 : 2);
    return gb / (ms / 5 * 1e-3);
}

int main()
{
    const long long W = 2455, R = 2056;
    const size_t total = (size_t)W * R * 512;
    void *A, *B; unsigned* sink;
    CK(hipMalloc(&A, total)); CK(hipMalloc(&B, total)); CK(hipMalloc(&sink, 4)); CK(hipMemset(A, 1, total)); CK(hipMemset(B, 1, total));
    printf("read         plain %6.0f   nt-load %6.0f GB/s\n", run<0, false, false>(A, B, W, R, sink), run<0, true, false>(A, B, W, R, sink));
    printf("read+write   plain %6.0f   nt-load %6.0f   nt-store %6.0f   both %6.0f GB/s\n", run<1, false, false>(A, B, W, R, sink),
           run<1, true, false>(A, B, W, R, sink), run<1, false, true>(A, B, W, R, sink), run<1, true, true>(A, B, W, R, sink));
    printf("rmw          plain %6.0f   nt-load %6.0f   nt-store %6.0f   both %6.0f GB/s\n", run<2, false, false>(A, B, W, R, sink),
           run<2, true, false>(A, B, W, R, sink), run<2, false, true>(A, B, W, R, sink), run<2, true, true>(A, B, W, R, sink));
    return 0;
}
