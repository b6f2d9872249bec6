// order.hip -- does vmcnt count down IN ORDER on gfx950 when loads and stores (global / buffer / scratch) are mixed?
// The compiler assumes it does: for a wait on an older load it emits  s_waitcnt vmcnt(N)  with N = the number of vector
// memory operations issued after that load, whatever their kind (SIInsertWaitcnts, targets without a separate store
// counter).  k_pair with a vector-loaded minima record read garbage through v_readlane behind exactly such a wait, and
// was correct with vmcnt(0) in its place.
//   hipcc --offload-arch=gfx950 -O3 scripts/micro/order.hip -o /tmp/order && /tmp/order
// Each wave: sentinel -> VGPR; a cold load into that VGPR (random line of a 2 GiB buffer); N younger operations of kind X
// to hot addresses; s_waitcnt vmcnt(N); copy the VGPR ("early"); s_waitcnt vmcnt(0); copy again ("late").  In-order
// completion means early == late always.  Counted: lanes where early is still the sentinel.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

constexpr uint32_t SENT = 0xDEADBEEFu;

__device__ __forceinline__ uint32_t rnd(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

// KIND 0: younger global stores   1: younger scratch stores   2: younger global loads (hot line)   3: younger buffer stores
// 4: younger scratch loads
template <int KIND>
__global__ void __launch_bounds__(256) k_order(const uint32_t* __restrict__ cold, size_t ncold, uint32_t* __restrict__ hot,
                                               unsigned long long* __restrict__ bad, int iters)
{
    volatile uint32_t priv[16];                              // forces a scratch allocation
    const uint32_t tid = blockIdx.x * 256 + threadIdx.x;
    for (int i = 0; i < 16; ++i) priv[i] = tid + i;
    unsigned long long nbad = 0;
    uint32_t* myhot = hot + (size_t)tid * 8;
    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
    const uint64_t hb = (uint64_t)hot;                       // wave-uniform base; the per-lane part goes into the offset VGPR
    u4 hr = { (uint32_t)hb, (uint32_t)(hb >> 32) & 0xFFFFu, 0xFFFFFFFFu, 0x00020000u };
    hr.x = __builtin_amdgcn_readfirstlane(hr.x); hr.y = __builtin_amdgcn_readfirstlane(hr.y);
    const uint32_t hoff = tid * 32;
    for (int it = 0; it < iters; ++it) {
        const uint32_t* p = cold + ((size_t)rnd(tid * 977u + it * 7919u) % (ncold / 32)) * 32;   // a 128-byte line of its own
        uint32_t early, late, ld, t0, t1, t2, t3;
        const uint32_t val = tid ^ it;
        if constexpr (KIND == 0)
            asm volatile("v_mov_b32 %2, %5\n s_nop 4\n global_load_dword %2, %3, off\n"
                         "global_store_dword %4, %6, off\n global_store_dword %4, %6, off offset:4\n"
                         "global_store_dword %4, %6, off offset:8\n global_store_dword %4, %6, off offset:12\n"
                         "s_waitcnt vmcnt(4)\n v_mov_b32 %0, %2\n s_waitcnt vmcnt(0)\n v_mov_b32 %1, %2\n"
                         : "=&v"(early), "=&v"(late), "=&v"(ld) : "v"(p), "v"(myhot), "s"(SENT), "v"(val) : "memory");
        else if constexpr (KIND == 1)
            asm volatile("v_mov_b32 %2, %4\n s_nop 4\n global_load_dword %2, %3, off\n"
                         "scratch_store_dword off, %5, off offset:0\n scratch_store_dword off, %5, off offset:4\n"
                         "scratch_store_dword off, %5, off offset:8\n scratch_store_dword off, %5, off offset:12\n"
                         "s_waitcnt vmcnt(4)\n v_mov_b32 %0, %2\n s_waitcnt vmcnt(0)\n v_mov_b32 %1, %2\n"
                         : "=&v"(early), "=&v"(late), "=&v"(ld) : "v"(p), "s"(SENT), "v"(val) : "memory");
        else if constexpr (KIND == 2)
            asm volatile("v_mov_b32 %2, %9\n s_nop 4\n global_load_dword %2, %7, off\n"
                         "global_load_dword %3, %8, off\n global_load_dword %4, %8, off offset:4\n"
                         "global_load_dword %5, %8, off offset:8\n global_load_dword %6, %8, off offset:12\n"
                         "s_waitcnt vmcnt(4)\n v_mov_b32 %0, %2\n s_waitcnt vmcnt(0)\n v_mov_b32 %1, %2\n"
                         : "=&v"(early), "=&v"(late), "=&v"(ld), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3) : "v"(p), "v"(myhot), "s"(SENT) : "memory");
        else if constexpr (KIND == 3)
            asm volatile("v_mov_b32 %2, %5\n s_nop 4\n global_load_dword %2, %3, off\n"
                         "buffer_store_dword %6, %7, %4, 0 offen\n buffer_store_dword %6, %7, %4, 0 offen offset:4\n"
                         "buffer_store_dword %6, %7, %4, 0 offen offset:8\n buffer_store_dword %6, %7, %4, 0 offen offset:12\n"
                         "s_waitcnt vmcnt(4)\n v_mov_b32 %0, %2\n s_waitcnt vmcnt(0)\n v_mov_b32 %1, %2\n"
                         : "=&v"(early), "=&v"(late), "=&v"(ld) : "v"(p), "s"(hr), "s"(SENT), "v"(val), "v"(hoff) : "memory");
        else
            asm volatile("v_mov_b32 %2, %8\n s_nop 4\n global_load_dword %2, %7, off\n"
                         "scratch_load_dword %3, off, off offset:0\n scratch_load_dword %4, off, off offset:4\n"
                         "scratch_load_dword %5, off, off offset:8\n scratch_load_dword %6, off, off offset:12\n"
                         "s_waitcnt vmcnt(4)\n v_mov_b32 %0, %2\n s_waitcnt vmcnt(0)\n v_mov_b32 %1, %2\n"
                         : "=&v"(early), "=&v"(late), "=&v"(ld), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3) : "v"(p), "s"(SENT) : "memory");
        if (early != late) ++nbad;
    }
    if (nbad) atomicAdd(bad, nbad);
    if (priv[tid & 15] == 0x12345) hot[0] = 1;
}

int main()
{
    const size_t ncold = (size_t)1 << 29;                   // 2 GiB of dwords
    uint32_t *cold, *hot;
    unsigned long long* bad;
    hipMalloc(&cold, ncold * 4);
    hipMalloc(&hot, (size_t)1024 * 256 * 8 * 4);
    hipMalloc(&bad, 8);
    hipMemset(cold, 0x5A, ncold * 4);                       // every dword 0x5A5A5A5A: never the sentinel
    hipMemset(hot, 0, (size_t)1024 * 256 * 8 * 4);
    const char* names[] = { "global stores", "scratch stores", "global loads (hot)", "buffer stores", "scratch loads" };
    for (int kind = 0; kind < 5; ++kind) {
        hipMemset(bad, 0, 8);
        const int iters = 2000;
        switch (kind) {
            case 0: hipLaunchKernelGGL((k_order<0>), dim3(1024), dim3(256), 0, 0, cold, ncold, hot, bad, iters); break;
            case 1: hipLaunchKernelGGL((k_order<1>), dim3(1024), dim3(256), 0, 0, cold, ncold, hot, bad, iters); break;
            case 2: hipLaunchKernelGGL((k_order<2>), dim3(1024), dim3(256), 0, 0, cold, ncold, hot, bad, iters); break;
            case 3: hipLaunchKernelGGL((k_order<3>), dim3(1024), dim3(256), 0, 0, cold, ncold, hot, bad, iters); break;
            case 4: hipLaunchKernelGGL((k_order<4>), dim3(1024), dim3(256), 0, 0, cold, ncold, hot, bad, iters); break;
        }
        unsigned long long h = 0;
        hipError_t e = hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost);
        printf("cold load, then 4 younger %-20s, s_waitcnt vmcnt(4): %llu of %llu lane-reads saw the register before the load landed (%s)\n",
               names[kind], h, 1024ull * 256 * iters, hipGetErrorString(e));
    }
    return 0;
}
