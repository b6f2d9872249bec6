// wr.hip -- what does a WRITE stream cost on MI355X?  The aggregation's timeline fits  t = R / 7.96 TB/s + W / 2.84 TB/s
// (DESIGN.md 4.1): writes of S are 2.8x as expensive per byte as reads.  Is that the store form (8 B per lane, nt), the
// access pattern (512-byte vectors of strided chains) or the part?
//   hipcc --offload-arch=gfx950 -O3 scripts/micro/wr.hip -o /tmp/wr && /tmp/wr
// Patterns over a 2.5 GiB volume [rows][cols][128 dwords] (config B: 2058 x 2455 x 512 B):
//   lin   : grid-stride, W bytes per lane, fully contiguous
//   col   : one wave per column chain walking rows (512-B vectors 1.26 MB apart), like k_pair on the column family
//   row   : one wave per row chain walking columns (contiguous 512-B vectors)
// Modes: wo = write only, ro = read only, rw = read + write of the SAME vector (S += ...), r2w = read two volumes, write one
// (what a middle pair kernel does: C, S -> S).  aux: 0 default, 2 nt, 1 sc0, 16 sc1, 17 sc0 sc1, 3 nt sc0.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

typedef uint32_t v2u __attribute__((__vector_size__(8)));
typedef uint32_t v4u __attribute__((__vector_size__(16)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ __forceinline__ rsrc_t mk(const void* p) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0xFFFFFFFF, 0x00020000); }

constexpr int ROWS = 2058, COLS = 2455, VD = 128;            // dwords per vector
constexpr size_t VOL = (size_t)ROWS * COLS * VD * 4;

// MODE 0 wo, 1 ro, 2 rw, 3 r2w.  PAT 0 col, 1 row.  One wave per chain, U vectors in flight.
template <int MODE, int PAT, int AUX, int U>
__global__ void __launch_bounds__(256) k_chain(const uint32_t* __restrict__ A, uint32_t* __restrict__ B, uint32_t* __restrict__ sink)
{
    const int lane = threadIdx.x & 63;
    const int c = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nch = PAT == 0 ? COLS : ROWS, n = PAT == 0 ? ROWS : COLS;
    if (c >= nch) return;
    const long long pix0 = PAT == 0 ? c : (long long)c * COLS, pstep = PAT == 0 ? COLS : 1;
    const uint32_t voff = lane * 8, sstep = (uint32_t)pstep * 512;
    v2u acc = { 0, 0 };
    for (int t = 0; t + U <= n; t += U) {
        const rsrc_t ra = mk(A + (pix0 + (long long)t * pstep) * VD), rb = mk(B + (pix0 + (long long)t * pstep) * VD);
        v2u a[U], b[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (MODE == 3) a[u] = __builtin_amdgcn_raw_buffer_load_b64(ra, voff, u * sstep, AUX);
            if (MODE >= 1) b[u] = __builtin_amdgcn_raw_buffer_load_b64(rb, voff, u * sstep, AUX);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            v2u v = { (uint32_t)t, (uint32_t)u };
            if (MODE >= 1) v += b[u];
            if (MODE == 3) v += a[u];
            if (MODE == 1) acc += v;
            else __builtin_amdgcn_raw_buffer_store_b64(v, rb, voff, u * sstep, AUX);
        }
    }
    if (MODE == 1 && acc[0] == 0x12345678u) sink[lane] = acc[1];
}

// grid-stride, 16 B per lane
template <int MODE, int AUX>
__global__ void __launch_bounds__(256) k_lin(const v4u* __restrict__ A, v4u* __restrict__ B, size_t n, v4u* __restrict__ sink)
{
    const rsrc_t ra = mk(A), rb = mk(B);
    v4u acc = { 0, 0, 0, 0 };
    // 4 GiB window of a buffer descriptor: volumes here are 2.5 GiB
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const uint32_t off = (uint32_t)(i * 16);
        v4u v = { (uint32_t)i, 1, 2, 3 };
        if (MODE >= 1) v += __builtin_amdgcn_raw_buffer_load_b128(rb, off, 0, AUX);
        if (MODE == 3) v += __builtin_amdgcn_raw_buffer_load_b128(ra, off, 0, AUX);
        if (MODE == 1) acc += v;
        else __builtin_amdgcn_raw_buffer_store_b128(v, rb, off, 0, AUX);
    }
    if (MODE == 1 && acc[0] == 0x12345678u) sink[0] = acc;
}

static hipEvent_t e0, e1;
template <typename F>
static double timeit(F&& f, int iters)
{
    f(); f();
    hipEventRecord(e0, 0);
    for (int i = 0; i < iters; ++i) f();
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / iters;
}

template <int PAT, int AUX, int U>
static void chain_row(const uint32_t* A, uint32_t* B, uint32_t* sink, const char* name)
{
    const int nch = PAT == 0 ? COLS : ROWS;
    const dim3 g((nch + 3) / 4), b(256);
    const double t0 = timeit([&] { hipLaunchKernelGGL((k_chain<0, PAT, AUX, U>), g, b, 0, 0, A, B, sink); }, 5);
    const double t1 = timeit([&] { hipLaunchKernelGGL((k_chain<1, PAT, AUX, U>), g, b, 0, 0, A, B, sink); }, 5);
    const double t2 = timeit([&] { hipLaunchKernelGGL((k_chain<2, PAT, AUX, U>), g, b, 0, 0, A, B, sink); }, 5);
    const double t3 = timeit([&] { hipLaunchKernelGGL((k_chain<3, PAT, AUX, U>), g, b, 0, 0, A, B, sink); }, 5);
    const double gb = VOL / 1e9;
    printf("%-22s aux %2d U %2d | wo %5.2f  ro %5.2f  rw %5.2f  r2w %5.2f TB/s | ms %6.3f %6.3f %6.3f %6.3f\n", name, AUX, U, gb / t0, gb / t1,
           2 * gb / t2, 3 * gb / t3, t0, t1, t2, t3);
}

template <int AUX>
static void lin_row(const v4u* A, v4u* B, v4u* sink)
{
    const size_t n = VOL / 16;
    const dim3 g(256 * 16), b(256);
    const double t0 = timeit([&] { hipLaunchKernelGGL((k_lin<0, AUX>), g, b, 0, 0, A, B, n, sink); }, 5);
    const double t1 = timeit([&] { hipLaunchKernelGGL((k_lin<1, AUX>), g, b, 0, 0, A, B, n, sink); }, 5);
    const double t2 = timeit([&] { hipLaunchKernelGGL((k_lin<2, AUX>), g, b, 0, 0, A, B, n, sink); }, 5);
    const double t3 = timeit([&] { hipLaunchKernelGGL((k_lin<3, AUX>), g, b, 0, 0, A, B, n, sink); }, 5);
    const double gb = VOL / 1e9;
    printf("%-22s aux %2d      | wo %5.2f  ro %5.2f  rw %5.2f  r2w %5.2f TB/s | ms %6.3f %6.3f %6.3f %6.3f\n", "lin 16B/lane", AUX, gb / t0, gb / t1,
           2 * gb / t2, 3 * gb / t3, t0, t1, t2, t3);
}

int main()
{
    uint32_t *A, *B, *sink;
    hipMalloc(&A, VOL + (1 << 20)); hipMalloc(&B, VOL + (1 << 20)); hipMalloc(&sink, 4096);
    hipMemset(A, 1, VOL); hipMemset(B, 2, VOL);
    hipEventCreate(&e0); hipEventCreate(&e1);
    {
        const double t = timeit([&] { hipMemsetAsync(B, 3, VOL, 0); }, 5);
        printf("hipMemsetAsync          %5.2f TB/s\n", VOL / 1e9 / t);
    }
    lin_row<0>((const v4u*)A, (v4u*)B, (v4u*)sink);
    lin_row<2>((const v4u*)A, (v4u*)B, (v4u*)sink);
    lin_row<1>((const v4u*)A, (v4u*)B, (v4u*)sink);
    lin_row<16>((const v4u*)A, (v4u*)B, (v4u*)sink);
    lin_row<17>((const v4u*)A, (v4u*)B, (v4u*)sink);
    lin_row<3>((const v4u*)A, (v4u*)B, (v4u*)sink);
    chain_row<0, 0, 8>(A, B, sink, "col chains 512B");
    chain_row<0, 2, 8>(A, B, sink, "col chains 512B");
    chain_row<0, 1, 8>(A, B, sink, "col chains 512B");
    chain_row<0, 16, 8>(A, B, sink, "col chains 512B");
    chain_row<0, 17, 8>(A, B, sink, "col chains 512B");
    chain_row<0, 2, 16>(A, B, sink, "col chains 512B");
    chain_row<1, 0, 8>(A, B, sink, "row chains 512B");
    chain_row<1, 2, 8>(A, B, sink, "row chains 512B");
    chain_row<1, 17, 8>(A, B, sink, "row chains 512B");
    chain_row<1, 2, 16>(A, B, sink, "row chains 512B");
    return 0;
}
