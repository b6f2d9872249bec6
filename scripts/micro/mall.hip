// micro-benchmark: does the 256 MiB Infinity Cache serve re-reads of a band of the cost volume faster than HBM?
// Access pattern of the chain kernels: one wave per column, 512-byte vectors, stride = one image row (W vectors).
// For a band of R rows (R * W * 512 B) the same band is (a) read repeatedly, (b) written by one kernel and read by
// the next, (c) read-modify-written repeatedly.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int MODE>   // 0 read, 1 write, 2 read-modify-write
__global__ void __launch_bounds__(256) k_band(uint2* __restrict__ A, long long W, long long R, unsigned* sink)
{
    const int lane = threadIdx.x & 63;
    const long long c = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (c >= W) return;
    uint2* a = A + c * 64 + lane;
    unsigned acc = 0;
    for (long long k0 = 0; k0 + 8 <= R; k0 += 8) {
        uint2 v[8];
        if (MODE != 1) {
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = a[(k0 + u) * W * 64];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (MODE == 0) acc += v[u].x;
            else if (MODE == 1) a[(k0 + u) * W * 64] = make_uint2((unsigned)k0, lane);
            else { uint2 o = v[u]; o.x += 1; a[(k0 + u) * W * 64] = o; }
        }
    }
    if (MODE == 0 && acc == 0xdeadbeef) *sink = acc;
}

int main()
{
    const long long W = 2455;
    const size_t total = (size_t)W * 2058 * 512;
    void* A; unsigned* sink;
    CK(hipMalloc(&A, total)); CK(hipMalloc(&sink, 4)); CK(hipMemset(A, 1, total));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const dim3 grid((unsigned)((W + 3) / 4)), block(256);
    printf("band rows | MiB   | re-read GB/s | write->read GB/s (read leg) | rmw GB/s (r+w bytes)\n");
    for (long long R : { 16LL, 32LL, 64LL, 96LL, 128LL, 160LL, 192LL, 256LL, 384LL, 512LL, 1024LL, 2056LL }) {
        const double mib = (double)R * W * 512 / (1 << 20), gb = (double)R * W * 512 / 1e9;
        const int reps = 20;
        float ms;
        // (a) repeated reads
        hipLaunchKernelGGL(k_band<0>, grid, block, 0, 0, (uint2*)A, W, R, sink);
        hipEventRecord(e0);
        for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k_band<0>, grid, block, 0, 0, (uint2*)A, W, R, sink);
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        const double rr = gb / (ms / reps * 1e-3);
        // (b) write then read: time the pair, subtract a write-only run
        float mw, mwr;
        hipEventRecord(e0);
        for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k_band<1>, grid, block, 0, 0, (uint2*)A, W, R, sink);
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&mw, e0, e1);
        hipEventRecord(e0);
        for (int r = 0; r < reps; ++r) {
            hipLaunchKernelGGL(k_band<1>, grid, block, 0, 0, (uint2*)A, W, R, sink);
            hipLaunchKernelGGL(k_band<0>, grid, block, 0, 0, (uint2*)A, W, R, sink);
        }
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&mwr, e0, e1);
        const double wr = gb / ((mwr - mw) / reps * 1e-3), wo = gb / (mw / reps * 1e-3);
        // (c) repeated read-modify-write
        hipEventRecord(e0);
        for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k_band<2>, grid, block, 0, 0, (uint2*)A, W, R, sink);
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        const double rmw = 2 * gb / (ms / reps * 1e-3);
        printf("%9lld | %5.0f | %12.0f | %10.0f (write-only %5.0f) | %8.0f   [%.1f us per read launch]\n", R, mib, rr, wr, wo, rmw,
               gb / rr * 1e6);
    }
    return 0;
}
