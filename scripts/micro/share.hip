// share.hip -- do two row-synchronous readers of the SAME volume cost less than readers of two different volumes
// (second touch served by the 256 MiB Infinity Cache / L2)?  Column-chain pattern of wr.hip (one wave per column walking
// rows, 512-B vectors), readers launched concurrently on two streams, or fused in one kernel (each wave reads its
// column of volume A and the column (c + shift) of volume A or B).
//   hipcc --offload-arch=gfx950 -O3 scripts/micro/share.hip -o /tmp/share && /tmp/share
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t v2u __attribute__((__vector_size__(8)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ __forceinline__ rsrc_t mk(const void* p) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0xFFFFFFFF, 0x00020000); }
constexpr int ROWS = 2058, COLS = 2455, VD = 128;
constexpr size_t VOL = (size_t)ROWS * COLS * VD * 4;

template <int AUX, int U>
__global__ void __launch_bounds__(256) k_read(const uint32_t* __restrict__ A, int shift, int lag, uint32_t* __restrict__ sink)
{
    const int lane = threadIdx.x & 63;
    int c = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (c >= COLS) return;
    c = (c + shift) % COLS;
    const uint32_t voff = lane * 8, sstep = COLS * 512;
    v2u acc = { 0, 0 };
    for (int t = 0; t + U <= ROWS; t += U) {
        const rsrc_t ra = mk(A + ((long long)t * COLS + c) * VD);
        v2u a[U];
#pragma unroll
        for (int u = 0; u < U; ++u) a[u] = __builtin_amdgcn_raw_buffer_load_b64(ra, voff, u * sstep, AUX);
#pragma unroll
        for (int u = 0; u < U; ++u) acc += a[u];
    }
    if (acc[0] == 0x12345678u) sink[lane] = acc[1];
}
// fused: each wave reads column c of A and column (c + shift) of B in the same row
template <int AUX, int U>
__global__ void __launch_bounds__(256) k_read2(const uint32_t* __restrict__ A, const uint32_t* __restrict__ B, int shift, uint32_t* __restrict__ sink)
{
    const int lane = threadIdx.x & 63;
    const int c = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (c >= COLS) return;
    const int c2 = (c + shift) % COLS;
    const uint32_t voff = lane * 8, sstep = COLS * 512;
    v2u acc = { 0, 0 };
    for (int t = 0; t + U <= ROWS; t += U) {
        const rsrc_t ra = mk(A + ((long long)t * COLS + c) * VD), rb = mk(B + ((long long)t * COLS + c2) * VD);
        v2u a[U], b[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { a[u] = __builtin_amdgcn_raw_buffer_load_b64(ra, voff, u * sstep, AUX); b[u] = __builtin_amdgcn_raw_buffer_load_b64(rb, voff, u * sstep, AUX); }
#pragma unroll
        for (int u = 0; u < U; ++u) acc += a[u] + b[u];
    }
    if (acc[0] == 0x12345678u) sink[lane] = acc[1];
}
int main()
{
    uint32_t *A, *B, *sink;
    (void)hipMalloc(&A, VOL + (1 << 20)); (void)hipMalloc(&B, VOL + (1 << 20)); (void)hipMalloc(&sink, 4096);
    (void)hipMemset(A, 1, VOL); (void)hipMemset(B, 2, VOL);
    hipStream_t s1, s2; (void)hipStreamCreate(&s1); (void)hipStreamCreate(&s2);
    hipEvent_t e0, e1, e2; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1); (void)hipEventCreate(&e2);
    const dim3 g((COLS + 3) / 4), b(256);
    auto run = [&](const char* name, auto&& f, double gb) {
        float best = 1e9;
        for (int it = 0; it < 6; ++it) {
            (void)hipDeviceSynchronize();
            (void)hipEventRecord(e0, s1);
            (void)hipStreamWaitEvent(s2, e0, 0);
            f();
            (void)hipEventRecord(e2, s2);
            (void)hipStreamWaitEvent(s1, e2, 0);
            (void)hipEventRecord(e1, s1);
            (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            if (it > 0 && ms < best) best = ms;
        }
        printf("%-44s %7.3f ms  %5.2f TB/s (bytes touched)\n", name, best, gb / best);
    };
    const double gb = VOL / 1e9;
    run("one reader (nt)", [&] { hipLaunchKernelGGL((k_read<2, 8>), g, b, 0, s1, A, 0, 0, sink); }, gb);
    run("one reader (default policy)", [&] { hipLaunchKernelGGL((k_read<0, 8>), g, b, 0, s1, A, 0, 0, sink); }, gb);
    run("2 streams, different volumes (nt)", [&] { hipLaunchKernelGGL((k_read<2, 8>), g, b, 0, s1, A, 0, 0, sink); hipLaunchKernelGGL((k_read<2, 8>), g, b, 0, s2, B, 0, 0, sink); }, 2 * gb);
    run("2 streams, same volume (nt)", [&] { hipLaunchKernelGGL((k_read<2, 8>), g, b, 0, s1, A, 0, 0, sink); hipLaunchKernelGGL((k_read<2, 8>), g, b, 0, s2, A, 1000, 0, sink); }, 2 * gb);
    run("2 streams, different volumes (default)", [&] { hipLaunchKernelGGL((k_read<0, 8>), g, b, 0, s1, A, 0, 0, sink); hipLaunchKernelGGL((k_read<0, 8>), g, b, 0, s2, B, 0, 0, sink); }, 2 * gb);
    run("2 streams, same volume (default)", [&] { hipLaunchKernelGGL((k_read<0, 8>), g, b, 0, s1, A, 0, 0, sink); hipLaunchKernelGGL((k_read<0, 8>), g, b, 0, s2, A, 1000, 0, sink); }, 2 * gb);
    run("fused, different volumes (nt)", [&] { hipLaunchKernelGGL((k_read2<2, 4>), g, b, 0, s1, A, B, 1000, sink); }, 2 * gb);
    run("fused, same volume shifted 1000 cols (nt)", [&] { hipLaunchKernelGGL((k_read2<2, 4>), g, b, 0, s1, A, A, 1000, sink); }, 2 * gb);
    run("fused, different volumes (default)", [&] { hipLaunchKernelGGL((k_read2<0, 4>), g, b, 0, s1, A, B, 1000, sink); }, 2 * gb);
    run("fused, same volume shifted 1000 cols (default)", [&] { hipLaunchKernelGGL((k_read2<0, 4>), g, b, 0, s1, A, A, 1000, sink); }, 2 * gb);
    run("fused, same volume shifted 4 cols (default)", [&] { hipLaunchKernelGGL((k_read2<0, 4>), g, b, 0, s1, A, A, 4, sink); }, 2 * gb);
    run("fused, same volume shifted 32 cols (default)", [&] { hipLaunchKernelGGL((k_read2<0, 4>), g, b, 0, s1, A, A, 32, sink); }, 2 * gb);
    return 0;
}
