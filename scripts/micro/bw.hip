// micro-benchmark: achievable HBM bandwidth for the access patterns of the chain kernels (8 B per lane, 512-B vectors)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// each wave walks a chain of n vectors (uint2 per lane), stride in vectors; MODE 0 read-only, 1 read+write (other volume), 2 RMW same
template <int U, int MODE, typename VT>
__global__ void __launch_bounds__(256) k_chain(const VT* __restrict__ A, VT* __restrict__ B, long long nchains, long long n,
                                               long long chain_stride, long long step_stride, unsigned* sink)
{
    const int lane = threadIdx.x & 63;
    const long long c = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (c >= nchains) return;
    const VT* a = A + (c * chain_stride) * 64 + lane;
    VT* b = B + (c * chain_stride) * 64 + lane;
    unsigned acc = 0;
    for (long long k0 = 0; k0 + U <= n; k0 += U) {
        VT v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = a[(k0 + u) * step_stride * 64];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (MODE == 0) acc += v[u].x;
            else { VT o = v[u]; o.x += 1; b[(k0 + u) * step_stride * 64] = o; }
        }
    }
    if (MODE == 0 && acc == 0xdeadbeef) *sink = acc;
}

template <int U, int MODE, typename VT>
float run(const void* A, void* B, long long nchains, long long n, long long cs, long long ss, unsigned* sink, int reps)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    dim3 grid((unsigned)((nchains + 3) / 4));
    hipLaunchKernelGGL((k_chain<U, MODE, VT>), grid, dim3(256), 0, 0, (const VT*)A, (VT*)B, nchains, n, cs, ss, sink);
    hipEventRecord(e0);
    for (int r = 0; r < reps; ++r)
        hipLaunchKernelGGL((k_chain<U, MODE, VT>), grid, dim3(256), 0, 0, (const VT*)A, (VT*)B, nchains, n, cs, ss, sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / reps;
}

int main()
{
    const long long W = 2455, H = 2058;            // vectors of 512 B (uint2 x 64 lanes)
    const size_t bytes = (size_t)W * H * 512;
    void *A, *B; unsigned* sink;
    CK(hipMalloc(&A, bytes * 2)); CK(hipMalloc(&B, bytes * 2)); CK(hipMalloc(&sink, 4));
    CK(hipMemset(A, 1, bytes * 2)); CK(hipMemset(B, 0, bytes * 2));
    const double gb = bytes / 1e9;
    struct { const char* name; long long nch, n, cs, ss; } pat[] = {
        { "rows   (chain=row, step=+1)     ", H, W, W, 1 },
        { "cols   (chain=col, step=+W)     ", W, H, 1, W },
        { "blocks (64-row column segments) ", W * 32, 64, 0, 0 },   // filled below
    };
    for (int p = 0; p < 2; ++p) {
        float r0 = run<8, 0, uint2>(A, B, pat[p].nch, pat[p].n, pat[p].cs, pat[p].ss, sink, 5);
        float r1 = run<8, 1, uint2>(A, B, pat[p].nch, pat[p].n, pat[p].cs, pat[p].ss, sink, 5);
        float r2 = run<8, 2, uint2>(A, A, pat[p].nch, pat[p].n, pat[p].cs, pat[p].ss, sink, 5);
        float r3 = run<16, 0, uint2>(A, B, pat[p].nch, pat[p].n, pat[p].cs, pat[p].ss, sink, 5);
        printf("%s 8B/lane: read %.3f ms (%.2f TB/s) | U16 read %.3f ms (%.2f TB/s) | copy %.3f ms (%.2f TB/s) | rmw %.3f ms (%.2f TB/s)\n",
               pat[p].name, r0, gb / r0, r3, gb / r3, r1, 2 * gb / r1, r2, 2 * gb / r2);
    }
    // 16 B per lane: vectors of 1 KB (config-E-like), same total bytes
    {
        const long long W2 = W / 2;
        float r0 = run<8, 0, uint4>(A, B, H, W2, W2, 1, sink, 5);
        float r1 = run<8, 1, uint4>(A, B, H, W2, W2, 1, sink, 5);
        float c0 = run<8, 0, uint4>(A, B, W2, H, 1, W2, sink, 5);
        float c1 = run<8, 1, uint4>(A, B, W2, H, 1, W2, sink, 5);
        const double g2 = (double)W2 * H * 1024 / 1e9;
        printf("16B/lane rows: read %.3f ms (%.2f TB/s) copy %.3f (%.2f TB/s) | cols: read %.3f (%.2f TB/s) copy %.3f (%.2f TB/s)\n",
               r0, g2 / r0, r1, 2 * g2 / r1, c0, g2 / c0, c1, 2 * g2 / c1);
    }
    // plain grid-stride streaming copy for reference
    return 0;
}
