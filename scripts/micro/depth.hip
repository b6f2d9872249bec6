// micro-benchmark: column-walk read bandwidth vs loads in flight per wave (U) and waves per column chain split (S row
// segments per column, i.e. S times more waves), 512-byte vectors, 2455 columns x 2056 rows, nt loads
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int U>
__global__ void __launch_bounds__(256) k_cols(const uint2* __restrict__ A, long long W, long long R, int S, unsigned* sink)
{
    const int lane = threadIdx.x & 63;
    const long long id = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (id >= W * S) return;
    const long long c = id % W, seg = id / W, rows = R / S;
    const uint2* a = A + (seg * rows * W + c) * 64 + lane;
    unsigned acc = 0;
    for (long long k0 = 0; k0 + U <= rows; k0 += U) {
        uint2 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { const uint2* p = a + (k0 + u) * W * 64; v[u].x = __builtin_nontemporal_load(&p->x); v[u].y = __builtin_nontemporal_load(&p->y); }
#pragma unroll
        for (int u = 0; u < U; ++u) acc += v[u].x ^ v[u].y;
    }
    if (acc == 0xdeadbeef) *sink = acc;
}

template <int U>
static double run(void* A, long long W, long long R, int S, unsigned* sink)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const dim3 grid((unsigned)((W * S + 3) / 4)), block(256);
    hipLaunchKernelGGL((k_cols<U>), grid, block, 0, 0, (const uint2*)A, W, R, S, sink);
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((k_cols<U>), grid, block, 0, 0, (const uint2*)A, W, R, S, sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return (double)R * W * 512 / 1e9 / (ms / 5 * 1e-3);
}

int main()
{
    const long long W = 2455, R = 2048;
    void* A; unsigned* sink;
    CK(hipMalloc(&A, (size_t)W * R * 512)); CK(hipMalloc(&sink, 4)); CK(hipMemset(A, 1, (size_t)W * R * 512));
    for (int S : { 1, 2, 4, 8 })
        printf("%d segment(s) per column (%5lld waves):  U=4 %6.0f   U=8 %6.0f   U=16 %6.0f   U=32 %6.0f GB/s\n", S, W * S,
               run<4>(A, W, R, S, sink), run<8>(A, W, R, S, sink), run<16>(A, W, R, S, sink), run<32>(A, W, R, S, sink));
    return 0;
}
