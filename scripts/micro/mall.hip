// mall.hip -- how fast is a read-modify-write stream (the S volume's access pattern) when its working set fits the
// 256 MiB Infinity Cache?  Repeated passes over a buffer of X MiB: rw (16 B load + 16 B store per lane) and read-only.
//   hipcc --offload-arch=gfx950 -O3 scripts/micro/mall.hip -o /tmp/mall && /tmp/mall
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <bool NT>
__global__ void __launch_bounds__(256) k_rw(u32x4* __restrict__ p, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        u32x4 v = NT ? __builtin_nontemporal_load(p + i) : p[i];
        v += 1u;
        if (NT) __builtin_nontemporal_store(v, p + i); else p[i] = v;
    }
}
template <bool NT>
__global__ void __launch_bounds__(256) k_ro(const u32x4* __restrict__ p, size_t n, u32x4* __restrict__ sink)
{
    u32x4 acc = { 0, 0, 0, 0 };
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) acc += NT ? __builtin_nontemporal_load(p + i) : p[i];
    if (acc.x == 0x12345678u) sink[0] = acc;
}

int main()
{
    const size_t maxb = (size_t)4 << 30;
    u32x4* buf; u32x4* sink;
    hipMalloc(&buf, maxb); hipMalloc(&sink, 64);
    hipMemset(buf, 0, maxb);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const size_t sizes_mb[] = { 8, 16, 32, 64, 96, 128, 160, 192, 256, 384, 512, 1024, 4096 };
    printf("%8s %12s %12s %12s %12s   (TB/s; rw counts read + write bytes)\n", "MiB", "rw", "rw_nt", "ro", "ro_nt");
    for (size_t mb : sizes_mb) {
        const size_t bytes = mb << 20, n = bytes / 16;
        const int iters = (int)((((size_t)16 << 30) / bytes) < 4 ? 4 : (((size_t)16 << 30) / bytes));
        const int grid = 256 * 8;
        float t[4];
        for (int mode = 0; mode < 4; ++mode) {
            auto launch = [&]() {
                switch (mode) {
                    case 0: hipLaunchKernelGGL(k_rw<false>, dim3(grid), dim3(256), 0, 0, buf, n); break;
                    case 1: hipLaunchKernelGGL(k_rw<true>, dim3(grid), dim3(256), 0, 0, buf, n); break;
                    case 2: hipLaunchKernelGGL(k_ro<false>, dim3(grid), dim3(256), 0, 0, buf, n, sink); break;
                    default: hipLaunchKernelGGL(k_ro<true>, dim3(grid), dim3(256), 0, 0, buf, n, sink); break;
                }
            };
            for (int w = 0; w < 3; ++w) launch();
            hipEventRecord(e0, 0);
            for (int it = 0; it < iters; ++it) launch();
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            hipEventElapsedTime(&t[mode], e0, e1);
            t[mode] = (float)((double)bytes * iters * (mode < 2 ? 2 : 1) / (t[mode] * 1e-3) / 1e12);
        }
        printf("%8zu %12.2f %12.2f %12.2f %12.2f\n", mb, t[0], t[1], t[2], t[3]);
    }
    return 0;
}
