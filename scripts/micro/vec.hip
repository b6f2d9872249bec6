// micro-benchmark: achieved bandwidth of the column-walk pattern as a function of the per-wave vector size
// (VB bytes per pixel vector = 64 lanes x VB/64 bytes); total bytes held constant (~2.5 GB), nt loads/stores
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int DW, int MODE>   // DW dwords per lane; MODE 0 read, 2 rmw
__global__ void __launch_bounds__(256) k_cols(uint32_t* __restrict__ A, long long W, long long R, unsigned* sink)
{
    const int lane = threadIdx.x & 63;
    const long long c = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (c >= W) return;
    uint32_t* a = A + (c * 64 + lane) * DW;
    unsigned acc = 0;
    for (long long k0 = 0; k0 + 8 <= R; k0 += 8) {
        uint32_t v[8][DW];
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int j = 0; j < DW; ++j) v[u][j] = __builtin_nontemporal_load(a + (k0 + u) * W * 64 * DW + j);
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int j = 0; j < DW; ++j) {
                if (MODE == 0) acc += v[u][j];
                else __builtin_nontemporal_store(v[u][j] + 1, a + (k0 + u) * W * 64 * DW + j);
            }
    }
    if (MODE == 0 && acc == 0xdeadbeef) *sink = acc;
}

template <int DW, int MODE>
static double run(void* A, long long W, long long R, unsigned* sink)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const dim3 grid((unsigned)((W + 3) / 4)), block(256);
    hipLaunchKernelGGL((k_cols<DW, MODE>), grid, block, 0, 0, (uint32_t*)A, W, R, sink);
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((k_cols<DW, MODE>), grid, block, 0, 0, (uint32_t*)A, W, R, sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return (double)R * W * 256 * DW / 1e9 * (MODE == 0 ? 1 : 2) / (ms / 5 * 1e-3);
}

int main()
{
    const size_t total = (size_t)2455 * 2056 * 512;
    void* A; unsigned* sink;
    CK(hipMalloc(&A, total + (1 << 20))); CK(hipMalloc(&sink, 4)); CK(hipMemset(A, 1, total));
    // same image height, width scaled so that the volume stays ~2.5 GB
    printf("vector  256 B: read %6.0f  rmw %6.0f GB/s\n", run<1, 0>(A, 4910, 2056, sink), run<1, 2>(A, 4910, 2056, sink));
    printf("vector  512 B: read %6.0f  rmw %6.0f GB/s\n", run<2, 0>(A, 2455, 2056, sink), run<2, 2>(A, 2455, 2056, sink));
    printf("vector 1024 B: read %6.0f  rmw %6.0f GB/s\n", run<4, 0>(A, 1227, 2056, sink), run<4, 2>(A, 1227, 2056, sink));
    printf("vector 1280 B: read %6.0f  rmw %6.0f GB/s\n", run<5, 0>(A, 982, 2056, sink), run<5, 2>(A, 982, 2056, sink));
    printf("vector 2048 B: read %6.0f  rmw %6.0f GB/s\n", run<8, 0>(A, 613, 2056, sink), run<8, 2>(A, 613, 2056, sink));
    return 0;
}
