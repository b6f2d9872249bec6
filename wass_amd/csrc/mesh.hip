// mesh.hip -- triangulation and the PovMesh stages on the device
// (SURVEY.md section 8 rows a10-a20).
//
// The reference keeps the organised cloud as a 40-byte AoS
// (wass_stereo/PovMesh.h:33-51); here it is structure-of-arrays in HBM
// (valid u8, x/y/z f64, gray u8) so that every pass streams only what it
// needs.  All geometry is fp64 and compiled with -ffp-contract=off: the same
// operations in the same order as the reference's plain x86-64 build, so the
// inlier counts of RANSAC / crop_plane are reproduced exactly.
#include "common.h"
#include "fmt_g6.h"

#include <math.h>
#include <stdlib.h>
#include <mutex>
#include <new>
#include <vector>

namespace wass {

struct GeomDev {
    double Kl[9], Kr[9], R[9], T[3];
    int use_custom;
    double R1[9], R2[9], P1[12], P2[12], HLi[9], HRi[9];
    double comp_over_scale;
};

__device__ __forceinline__ void mulv(const double* M, const double* v, double* o)
{
    o[0] = M[0] * v[0] + M[1] * v[1] + M[2] * v[2];
    o[1] = M[3] * v[0] + M[4] * v[1] + M[5] * v[2];
    o[2] = M[6] * v[0] + M[7] * v[1] + M[8] * v[2];
}
__device__ __forceinline__ void tmulv(const double* M, const double* v, double* o)
{
    o[0] = M[0] * v[0] + M[3] * v[1] + M[6] * v[2];
    o[1] = M[1] * v[0] + M[4] * v[1] + M[7] * v[2];
    o[2] = M[2] * v[0] + M[5] * v[1] + M[8] * v[2];
}

// Counters are striped over NSLOT addresses (summed on the host): tens of thousands of waves hitting one
// address serialise at ~12 ns per atomic, which would dominate these otherwise trivial kernels.
constexpr int NSLOT = 64;
__device__ __forceinline__ int slot_of_block() { return (int)((blockIdx.x * 7u + blockIdx.y * 13u + (threadIdx.x >> 6)) & (NSLOT - 1)); }

// StereoMatchEnv::unrectify (wass_stereo.cpp:299-324)
__device__ __forceinline__ void unrectify(const GeomDev& g, double u, double v, bool left, double* out)
{
    if (g.use_custom) {
        const double in[3] = { u, v, 1.0 };
        double r[3];
        mulv(left ? g.HLi : g.HRi, in, r);
        out[0] = r[0] / r[2]; out[1] = r[1] / r[2];
    } else {
        const double* K = left ? g.Kl : g.Kr;
        const double* P = left ? g.P1 : g.P2;
        const double xyw[3] = { (u - P[2]) / P[0], (v - P[6]) / P[5], 1.0 };
        double r[3];
        tmulv(left ? g.R1 : g.R2, xyw, r);
        r[0] /= r[2]; r[1] /= r[2];
        out[0] = r[0] * K[0] + K[2];
        out[1] = r[1] * K[4] + K[5];
    }
}

// triangulate(p,q,R,T) (wass_lib/triangulate.hpp:26-72) with the closed-form 3x3 solve
__device__ __forceinline__ void triangulate_point(const double* p, const double* q, const double* R, const double* T,
                                                  double* out)
{
    double Af[12], Bf[4], A[9], b[3];
    Af[0] = -1.0; Af[1] = 0.0; Af[2] = p[0];
    Af[3] = 0.0; Af[4] = -1.0; Af[5] = p[1];
    Af[6] = q[0] * R[6] - R[0]; Af[7] = q[0] * R[7] - R[1]; Af[8] = q[0] * R[8] - R[2];
    Af[9] = q[1] * R[6] - R[3]; Af[10] = q[1] * R[7] - R[4]; Af[11] = q[1] * R[8] - R[5];
    Bf[0] = 0.0; Bf[1] = 0.0;
    Bf[2] = T[0] - T[2] * q[0];
    Bf[3] = T[1] - T[2] * q[1];
    A[0] = Af[0] * Af[0] + Af[3] * Af[3] + Af[6] * Af[6] + Af[9] * Af[9];
    A[1] = Af[0] * Af[1] + Af[3] * Af[4] + Af[10] * Af[9] + Af[6] * Af[7];
    A[2] = Af[0] * Af[2] + Af[3] * Af[5] + Af[11] * Af[9] + Af[6] * Af[8];
    A[3] = A[1];
    A[4] = Af[1] * Af[1] + Af[10] * Af[10] + Af[4] * Af[4] + Af[7] * Af[7];
    A[5] = Af[10] * Af[11] + Af[1] * Af[2] + Af[4] * Af[5] + Af[7] * Af[8];
    A[6] = A[2];
    A[7] = A[5];
    A[8] = Af[11] * Af[11] + Af[2] * Af[2] + Af[5] * Af[5] + Af[8] * Af[8];
    b[0] = Af[0] * Bf[0] + Af[3] * Bf[1] + Af[6] * Bf[2] + Af[9] * Bf[3];
    b[1] = Af[1] * Bf[0] + Af[10] * Bf[3] + Af[4] * Bf[1] + Af[7] * Bf[2];
    b[2] = Af[2] * Bf[0] + Af[11] * Bf[3] + Af[5] * Bf[1] + Af[8] * Bf[2];
    double det = A[0] * (A[4] * A[8] - A[5] * A[7]) - A[1] * (A[3] * A[8] - A[5] * A[6]) +
                 A[2] * (A[3] * A[7] - A[4] * A[6]);
    if (det != 0) {
        det = 1. / det;
        out[0] = det * (b[0] * (A[4] * A[8] - A[5] * A[7]) - A[1] * (b[1] * A[8] - A[5] * b[2]) +
                        A[2] * (b[1] * A[7] - A[4] * b[2]));
        out[1] = det * (A[0] * (b[1] * A[8] - A[5] * b[2]) - b[0] * (A[3] * A[8] - A[5] * A[6]) +
                        A[2] * (A[3] * b[2] - b[1] * A[6]));
        out[2] = det * (A[0] * (A[4] * b[2] - b[1] * A[7]) - A[1] * (A[3] * b[2] - b[1] * A[6]) +
                        b[0] * (A[3] * A[7] - A[4] * A[6]));
    } else {
        out[0] = out[1] = out[2] = 0.0;
    }
}

// triangulate(StereoMatchEnv&) (wass_stereo.cpp:1173-1365): one thread per ROI pixel
// At most WASS_TRI_WAVES waves per SIMD: in a sequence this kernel -- fp64 divisions and an acos per pixel -- is the first big
// kernel of a frame's tail and runs underneath the NEXT frame's k_hsum_q, which is issue-bound.  At full occupancy it took
// the horizontal sum from 0.59 ms (alone) to 0.83 ms; held to one wave per SIMD it takes three times as long itself, which
// nobody waits for, and the cost stage of the next frame 1.79 instead of 2.05 ms (frame period -1.7 .. 3 %, alternating
// builds on one box).  Holding ALL tail kernels down makes the tail the bottleneck (10.3 ms per frame at one wave, 8.8 at two).
#ifndef WASS_TRI_WAVES
#define WASS_TRI_WAVES 1
#endif
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, WASS_TRI_WAVES))) k_triangulate(const float* __restrict__ disp, int W, int H, int rlx, int rrx, int rry,
                                                     int mw, int mh, GeomDev g, const uint8_t* __restrict__ right_img,
                                                     int img_w, int img_h, const uint8_t* __restrict__ lmask,
                                                     const uint8_t* __restrict__ rmask, double min_angle, double bx0,
                                                     double by0, double bx1, double by1, double cam_distance,
                                                     uint8_t* __restrict__ valid, double* __restrict__ X,
                                                     double* __restrict__ Y, double* __restrict__ Z,
                                                     uint8_t* __restrict__ gray, uint8_t* __restrict__ codes,
                                                     unsigned long long* __restrict__ count)
{
    unsigned int found = 0;
    // a workgroup owns a 256-column strip and walks rows: no 64-bit division per pixel
    const int u = blockIdx.x * 256 + threadIdx.x;
    if (u < mw)
    for (int v = blockIdx.y; v < mh; v += gridDim.y) {
    const size_t idx = (size_t)v * mw + u;
    bool ok = false;
    double P[3] = { 0, 0, 0 };
    uint8_t gv = 0;
    // what the reference paints into dbg_R0 (low nibble) / dbg_R1 (high nibble) for this pixel (:1216-1338); later
    // assignments overwrite earlier ones exactly as there
    unsigned int c0 = WASS_CODE_NONE, c1 = WASS_CODE_NONE;
    const float dv = disp[idx];
    const int xr = rrx + u, yr = rry + v;
    if (dv > 1.0f) {                                                  // min_disp = 1 (:1100,1177)
        const float xl = (float)((float)(xr - rrx + rlx) - dv + g.comp_over_scale);   // :1180
        const float yl = (float)yr;
        if (!(xl < 0 || xl >= (float)W)) {                            // :1183
            double pi[2], qi[2];
            unrectify(g, (double)xl, (double)yl, true, pi);
            unrectify(g, (double)xr, (double)yr, false, qi);
            bool skip = false;
            c0 = c1 = WASS_CODE_GREY;
            if (pi[0] < 1 || pi[0] >= img_w - 1 || pi[1] < 1 || pi[1] >= img_h - 1 ||
                qi[0] < 1 || qi[0] >= img_w - 1 || qi[1] < 1 || qi[1] >= img_h - 1) {
                skip = true;                                          // :1223
                c0 = c1 = WASS_CODE_OUTSIDE_IMAGE;
            }
            const double p[2] = { (pi[0] - g.Kl[2]) / g.Kl[0], (pi[1] - g.Kl[5]) / g.Kl[4] };
            const double q[2] = { (qi[0] - g.Kr[2]) / g.Kr[0], (qi[1] - g.Kr[5]) / g.Kr[4] };
            const bool inside = !skip;
            if (pi[0] <= bx0 || pi[1] <= by0 || pi[0] >= bx1 || pi[1] >= by1) { skip = true; c0 = c1 = WASS_CODE_OUTSIDE_BBOX; }   // :1236
            if (inside) {                                             // :1244,1250 (only where the reference's reads are in bounds)
                if (lmask && lmask[(size_t)(int)pi[1] * img_w + (int)pi[0]] == 0) { skip = true; c0 = WASS_CODE_OUTSIDE_BBOX; }
                if (rmask && rmask[(size_t)(int)qi[1] * img_w + (int)qi[0]] == 0) { skip = true; c1 = WASS_CODE_OUTSIDE_BBOX; }
            }
            if (min_angle > 0) {                                      // :1258-1269
                double d1[3] = { p[0], p[1], 1.0 }, qq[3] = { q[0], q[1], 1.0 }, d2[3];
                mulv(g.R, qq, d2);
                d2[0] += g.T[0]; d2[1] += g.T[1]; d2[2] += g.T[2];
                const double n1 = sqrt(d1[0] * d1[0] + d1[1] * d1[1] + d1[2] * d1[2]);
                const double n2 = sqrt(d2[0] * d2[0] + d2[1] * d2[1] + d2[2] * d2[2]);
                d1[0] /= n1; d1[1] /= n1; d1[2] /= n1;
                d2[0] /= n2; d2[1] /= n2; d2[2] /= n2;
                const double ang = fabs(acos(d1[0] * d2[0] + d1[1] * d2[1] + d1[2] * d2[2]) * 57.29577951);
                if (ang < min_angle) { skip = true; c0 = c1 = WASS_CODE_ANGLE; }
            }
            if (!skip) {
                triangulate_point(p, q, g.R, g.T, P);                 // :1286
                const double dist = sqrt(P[0] * P[0] + P[1] * P[1] + P[2] * P[2]);
                if (dist < cam_distance / 10.0 || P[2] < 1.0) c0 = c1 = WASS_CODE_TOO_CLOSE;             // :1329
                else if (dist > cam_distance * 200.0 || P[2] > 1E30) c0 = c1 = WASS_CODE_TOO_DISTANT;   // :1335
                else {
                    ok = true;
                    gv = right_img[(size_t)(int)qi[1] * img_w + (int)qi[0]];   // :1342
                }
            }
        }
    }
    valid[idx] = ok ? 1 : 0;
    X[idx] = ok ? P[0] : 0.0; Y[idx] = ok ? P[1] : 0.0; Z[idx] = ok ? P[2] : 0.0;
    gray[idx] = gv;
    codes[idx] = (uint8_t)(c0 | (c1 << 4));
    found += ok ? 1u : 0u;
    }
    for (int o = 32; o > 0; o >>= 1) found += __shfl_down(found, o);
    if ((threadIdx.x & 63) == 0 && found) atomicAdd(count + slot_of_block(), (unsigned long long)found);
}

// ------------------------------------------------------------------ z-gap percentile (PovMesh.cpp:888-926)
// gaps[3*idx + k] = |z - z(neighbour k in the row above)| as the fp64 bit pattern, ~0 when absent.
__global__ void __launch_bounds__(256) k_zgaps(const uint8_t* __restrict__ valid, const double* __restrict__ Z, int w, int h,
                                               unsigned long long* __restrict__ gaps, unsigned long long* __restrict__ count)
{
    const int j = blockIdx.x * 256 + threadIdx.x;
    const int i = blockIdx.y;
    if (j >= w) return;
    const size_t c = (size_t)i * w + j;
    unsigned long long g[3] = { ~0ull, ~0ull, ~0ull };
    int n = 0;
    if (i >= 1 && j >= 1 && j < w - 1 && valid[c]) {
        const double z = Z[c];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const size_t nb = c - w - 1 + k;
            if (valid[nb]) { g[k] = (unsigned long long)__double_as_longlong(fabs(z - Z[nb])); ++n; }
        }
    }
    gaps[3 * c] = g[0]; gaps[3 * c + 1] = g[1]; gaps[3 * c + 2] = g[2];
    // wave-level count
    int s = n;
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o);
    if ((threadIdx.x & 63) == 0 && s) atomicAdd(count + slot_of_block(), (unsigned long long)s);
}

// one MSD radix-select pass: histogram of an 11-bit digit among keys whose higher bits equal `prefix`
__global__ void __launch_bounds__(256) k_radix_hist(const unsigned long long* __restrict__ keys, size_t n, int shift,
                                                    unsigned int mask, int hi_shift, unsigned long long prefix,
                                                    unsigned int* __restrict__ hist)
{
    __shared__ unsigned int lh[2048];
    for (int i = threadIdx.x; i < 2048; i += 256) lh[i] = 0;
    __syncthreads();
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const unsigned long long k = keys[i];
        if (k == ~0ull) continue;
        if (hi_shift < 64 && (k >> hi_shift) != prefix) continue;
        atomicAdd(&lh[(unsigned)(k >> shift) & mask], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2048; i += 256)
        if (lh[i]) atomicAdd(&hist[i], lh[i]);
}

// ------------------------------------------------------------------ connected components (PovMesh.cpp:929-987)
// Union-find over RASTER indices (coalesced, neighbours are near in memory).  Horizontal runs are
// labelled up front by a segmented scan, so the merge pass only stitches runs that touch vertically.
// The reference finds seeds in COLUMN-MAJOR order and keeps the first strictly largest component, so
// ties are decided by the smallest column-major index (u*h + v) of a component -- tracked per root.
__device__ __forceinline__ int uf_find(int* parent, int a)
{
    // path halving: every visited node is re-pointed at its grandparent.  Labels only ever move towards
    // smaller ancestors, so the unsynchronised writes always store a valid ancestor.
    int p = parent[a];
    if (p == a) return a;
    int g;
    while ((g = parent[p]) != p) { parent[a] = g; a = p; p = g; }
    return p;
}
__device__ __forceinline__ void uf_union(int* parent, int a, int b)
{
    for (;;) {
        a = uf_find(parent, a);
        b = uf_find(parent, b);
        if (a == b) return;
        if (a > b) { const int t = a; a = b; b = t; }
        const int old = atomicMin(&parent[b], a);
        if (old == b) return;
        b = old;
    }
}
// link between raster pixel i and its left neighbour (same row)
__device__ __forceinline__ bool hlink(const uint8_t* valid, const double* Z, int w, int i, double zgap)
{
    return (i % w) != 0 && valid[i] && valid[i - 1] && fabs(Z[i] - Z[i - 1]) < zgap;
}
// link between raster pixel i and the pixel below it
__device__ __forceinline__ bool vlink(const uint8_t* valid, const double* Z, int w, int n, int i, double zgap)
{
    return i + w < n && valid[i] && valid[i + w] && fabs(Z[i] - Z[i + w]) < zgap;
}
__global__ void __launch_bounds__(256) k_ccl_init(const uint8_t* __restrict__ valid, const double* __restrict__ Z, int w, int n,
                                                  const double* __restrict__ zgap_p, int* __restrict__ parent,
                                                  unsigned int* __restrict__ size, unsigned int* __restrict__ mincm,
                                                  unsigned long long* __restrict__ best)
{
    const double zgap = *zgap_p;
    __shared__ int wmax[4];
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i == 0) *best = 0;                                       // k_ccl_best (three launches later) takes the maximum into it
    const bool v = i < n && valid[i];
    // start of my horizontal run inside this block = last "break" at or before me
    int s = (threadIdx.x == 0 || !v || !hlink(valid, Z, w, i, zgap)) ? i : -1;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(s, o); if (lane >= o) s = max(s, t); }
    if (lane == 63) wmax[wv] = s;
    __syncthreads();
    for (int k = 0; k < wv; ++k) s = max(s, wmax[k]);
    if (i < n) { parent[i] = v ? s : -1; size[i] = 0; mincm[i] = 0xFFFFFFFFu; }
}
__global__ void __launch_bounds__(256) k_ccl_merge(const uint8_t* __restrict__ valid, const double* __restrict__ Z, int w, int n,
                                                   const double* __restrict__ zgap_p, int* parent)
{
    const double zgap = *zgap_p;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n || !valid[i]) return;
    // runs were cut at block boundaries: stitch them
    if (threadIdx.x == 0 && hlink(valid, Z, w, i, zgap)) uf_union(parent, i, i - 1);
    if (vlink(valid, Z, w, n, i, zgap)) {
        // the pixel to the left already made this union if both rows continue their runs and it links down too
        const bool dup = hlink(valid, Z, w, i, zgap) && hlink(valid, Z, w, i + w, zgap) && vlink(valid, Z, w, n, i - 1, zgap);
        if (!dup) uf_union(parent, i, i + w);
    }
}
// every node points straight at its root.  Roots are fixed once merging is done; each thread writes only its
// own entry (a read-only walk), so no concurrent path-halving write can leave a node short of its root.
__global__ void __launch_bounds__(256) k_ccl_flatten(int n, int* parent)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    int p = parent[i];
    if (p < 0) return;
    int a = i;
    while (p != a) { a = p; p = parent[a]; }
    parent[i] = a;
}
// component sizes and smallest column-major index: one update per run of equal roots inside a wave, and the
// dominant root of a block is pre-aggregated in LDS
constexpr int CCL_COUNT_CHUNKS = 8;
__global__ void __launch_bounds__(1024) k_ccl_count(int n, int w, int h, const int* __restrict__ parent,
                                                    unsigned int* __restrict__ size, unsigned int* __restrict__ mincm)
{
    __shared__ int r0s;
    __shared__ unsigned int agg, aggmin;
    // a block walks CCL_COUNT_CHUNKS consecutive chunks of 1024 pixels: the root of its first pixel (almost always the
    // one big component) is totalled in LDS and reaches its global counter once per block -- updates of one address
    // serialise, and 5 000 blocks updating the same root were most of this kernel's time
    const int first = blockIdx.x * (1024 * CCL_COUNT_CHUNKS);
    if (threadIdx.x == 0) { r0s = first < n ? parent[first] : -1; agg = 0; aggmin = 0xFFFFFFFFu; }
    __syncthreads();
    const int r0 = r0s;
    const int lane = threadIdx.x & 63;
    for (int ch = 0; ch < CCL_COUNT_CHUNKS; ++ch) {
        const int i = first + ch * 1024 + threadIdx.x;
        const int r = i < n ? parent[i] : -1;
        const int prev = __shfl_up(r, 1);
        // a run also ends at a row boundary so that its head has the smallest column index of the run
        const bool head = r >= 0 && (lane == 0 || prev != r || (i % w) == 0);
        const unsigned long long heads = __ballot(head || r < 0);
        if (head) {
            const unsigned long long above = lane == 63 ? 0ull : (heads >> (lane + 1));
            const int len = above ? (__ffsll((long long)above)) : (64 - lane);
            const unsigned int cm = (unsigned)((i % w) * h + (i / w));
            if (r == r0) { atomicAdd(&agg, (unsigned)len); atomicMin(&aggmin, cm); }      // LDS
            else { atomicAdd(&size[r], (unsigned)len); atomicMin(&mincm[r], cm); }
        }
    }
    __syncthreads();
    if (threadIdx.x == 0 && agg) { atomicAdd(&size[r0], agg); atomicMin(&mincm[r0], aggmin); }
}
// best = max over roots of (size << 32 | ~mincm): largest size, then earliest column-major seed
__global__ void __launch_bounds__(256) k_ccl_best(int n, const int* __restrict__ parent, const unsigned int* __restrict__ size,
                                                  const unsigned int* __restrict__ mincm, unsigned long long* __restrict__ best)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    unsigned long long key = 0;
    if (i < n && parent[i] == i) key = ((unsigned long long)size[i] << 32) | (unsigned long long)(0xFFFFFFFFu - mincm[i]);
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long other = __shfl_down(key, o);
        key = other > key ? other : key;
    }
    if ((threadIdx.x & 63) == 0 && key) atomicMax(best, key);
}
// keep the component whose (size, mincm) equals the winner
__global__ void __launch_bounds__(256) k_ccl_keep(uint8_t* __restrict__ valid, int n, const int* __restrict__ parent,
                                                  const unsigned int* __restrict__ size, const unsigned int* __restrict__ mincm,
                                                  const unsigned long long* __restrict__ best)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int r = parent[i];
    bool keep = false;
    if (r >= 0) {
        const unsigned long long key = ((unsigned long long)size[r] << 32) | (unsigned long long)(0xFFFFFFFFu - mincm[r]);
        keep = key == *best;
    }
    valid[i] = keep ? 1 : 0;
}

// ------------------------------------------------------------------ device-resident scalars
// Decisions that only need a few numbers (the digit of a radix-select pass, the best RANSAC candidate, a 3x3
// eigenvector) are taken by single-workgroup kernels writing into this record, so that a whole stage chain runs
// without host round trips; the host reads the record once at the end.
struct DevState {
    unsigned long long sel_k, sel_prefix, sel_total;      // radix select: remaining rank, key prefix, number of keys
    int sel_hi_shift, sel_fail;
    double zgap;
    unsigned long long ccl_best;                          // (size << 32) | ~mincm of the winning component
    int ransac_found, pad0;
    unsigned long long ransac_best;
    double ransac_plane[4];
    double wsum, centroid[3], ninl;
    double plane[4];                                      // refined plane
    int refine_ok, pad1;
    // host-sync-free frame tail (wass_mesh_finish_frame_async): everything the xyzC encoder needs, decided on the device
    double rtR[9], rtT[3], rtRinv[9], rtTinv[3];
    double mn[3], sc[3];
    unsigned int npts, have_plane;
    unsigned long long kept1, kept2;
    unsigned long long ntri;                              // valid points the triangulation produced
    unsigned int ninl_sel, inl_text_bad;                  // refinement inliers seen by the selection for plane_refinement_inliers.xyz;
    unsigned long long inl_text_bytes;                    // ... bytes of that file's text when the device formatted it, numbers it could not format
};

// z gaps computed on the fly (no gap array): histogram of one 11-bit digit of the fp64 bit patterns that match the
// prefix found so far.  Blocks own a 256-column strip and walk rows (no index divisions); their totals go to one of
// GAP_HIST_COPIES copies of the global histogram (same-address atomics serialise), which k_radix_pick adds up.
constexpr int GAP_HIST_COPIES = 4;
// 5 x 11 + 9 bits = 64.  (13-bit digits, five passes: measured SLOWER -- 42 + 23 us per pass against 26 + 11: the 32 KB
// histogram per workgroup costs occupancy and the pick kernel scans four times the bins.)
constexpr int GAP_BITS = 11, GAP_BINS = 1 << GAP_BITS, GAP_PASSES = 6;
__global__ void __launch_bounds__(256) k_gap_hist(const uint8_t* __restrict__ valid, const double* __restrict__ Z, int w, int h,
                                                  int shift, unsigned int mask, const DevState* __restrict__ ds,
                                                  unsigned int* __restrict__ hist)
{
    __shared__ unsigned int lh[GAP_BINS];
    for (int i = threadIdx.x; i < GAP_BINS; i += 256) lh[i] = 0;
    __syncthreads();
    const int hi_shift = ds->sel_hi_shift;
    const unsigned long long prefix = ds->sel_prefix;
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= 1 && j < w - 1)
        for (int i = 1 + blockIdx.y; i < h; i += gridDim.y) {
            // all loads first and unconditionally (rows i-1 and i exist, columns j-1 .. j+1 too): independent requests in flight
            const size_t c = (size_t)i * w + j, up = c - w;
            const uint8_t vc = valid[c], v0 = valid[up - 1], v1 = valid[up], v2 = valid[up + 1];
            const double z = Z[c], z0 = Z[up - 1], z1 = Z[up], z2 = Z[up + 1];
            if (!vc) continue;
            const double zn[3] = { z0, z1, z2 };
            const uint8_t vn[3] = { v0, v1, v2 };
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                if (!vn[k]) continue;
                const unsigned long long key = (unsigned long long)__double_as_longlong(fabs(z - zn[k]));
                if (hi_shift < 64 && (key >> hi_shift) != prefix) continue;
                atomicAdd(&lh[(unsigned)(key >> shift) & mask], 1u);
            }
        }
    __syncthreads();
    unsigned int* mine = hist + (size_t)((blockIdx.x + blockIdx.y) % GAP_HIST_COPIES) * GAP_BINS;
    for (int i = threadIdx.x; i < GAP_BINS; i += 256)
        if (lh[i]) atomicAdd(&mine[i], lh[i]);
}
// one workgroup: pick the bin that holds rank k, extend the prefix; pass 0 also derives k from the percentile
__global__ void __launch_bounds__(256) k_radix_pick(unsigned int* __restrict__ hist, int pass, int shift, int nbits,
                                                    double percentile, DevState* __restrict__ ds)
{
    __shared__ unsigned long long tot[256];
    const int nb = 1 << nbits, per = (nb + 255) / 256;
    for (int i = threadIdx.x; i < GAP_BINS; i += 256) {                                  // fold the copies into copy 0
        unsigned int t = hist[i];
        for (int cpy = 1; cpy < GAP_HIST_COPIES; ++cpy) t += hist[(size_t)cpy * GAP_BINS + i];
        hist[i] = t;
    }
    __syncthreads();
    unsigned long long s = 0;
    for (int b = threadIdx.x * per; b < min(nb, (threadIdx.x + 1) * per); ++b) s += hist[b];
    tot[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long k = ds->sel_k;
        if (pass == 0) {
            unsigned long long total = 0;
            for (int t = 0; t < 256; ++t) total += tot[t];
            ds->sel_total = total;
            k = (unsigned long long)floor(percentile / 100.0 * (double)total);      // PovMesh.cpp:924
            if (total && k >= total) k = total - 1;
            ds->sel_prefix = 0;
            ds->sel_fail = total == 0;
        }
        if (!ds->sel_fail) {
            int t = 0;
            for (; t < 256; ++t) { if (k < tot[t]) break; k -= tot[t]; }
            int b = t * per;
            const int be = min(nb, (t + 1) * per);
            for (; b < be; ++b) { if (k < hist[b]) break; k -= hist[b]; }
            if (t >= 256 || b >= be) ds->sel_fail = 2;
            else {
                ds->sel_k = k;
                ds->sel_prefix = (ds->sel_prefix << nbits) | (unsigned long long)b;
                ds->sel_hi_shift = shift;
                if (shift == 0) ds->zgap = __longlong_as_double((long long)ds->sel_prefix);
            }
        }
        if (ds->sel_fail == 1) ds->zgap = __longlong_as_double(0x7FF8000000000000ll);     // no gaps: NaN
    }
    __syncthreads();
    for (int i = threadIdx.x; i < GAP_BINS * GAP_HIST_COPIES; i += 256) hist[i] = 0;        // ready for the next pass
}

// ------------------------------------------------------------------ RANSAC (PovMesh.cpp:665-777)
struct PlaneCand { double n[3]; double d; int ok; int pad; };

__global__ void k_ransac_planes(const uint8_t* __restrict__ valid, const double* __restrict__ X, const double* __restrict__ Y,
                                const double* __restrict__ Z, int w, const int32_t* __restrict__ uv, int rounds,
                                PlaneCand* __restrict__ cand, unsigned long long* __restrict__ counts,
                                unsigned long long* __restrict__ zero, int nzero)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    // the score / crop kernels that follow accumulate into these: cleared here instead of by separate fill launches
    for (int k = r; k < nzero; k += gridDim.x * blockDim.x) zero[k] = 0;
    if (r >= rounds) return;
    counts[r] = 0;
    const int32_t* c = uv + (size_t)r * 6;
    const size_t i1 = (size_t)c[1] * w + c[0], i2 = (size_t)c[3] * w + c[2], i3 = (size_t)c[5] * w + c[4];
    PlaneCand pc;
    pc.ok = valid[i1] && valid[i2] && valid[i3];
    pc.pad = 0;
    pc.n[0] = pc.n[1] = pc.n[2] = pc.d = 0.0;
    if (pc.ok) {
        const double a[3] = { X[i2] - X[i1], Y[i2] - Y[i1], Z[i2] - Z[i1] };
        const double b[3] = { X[i3] - X[i1], Y[i3] - Y[i1], Z[i3] - Z[i1] };
        double n[3] = { a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0] };
        const double nn = sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
        n[0] = n[0] / nn; n[1] = n[1] / nn; n[2] = n[2] / nn;
        if (n[2] < 0) { n[0] = n[0] * -1.0; n[1] = n[1] * -1.0; n[2] = n[2] * -1.0; }
        pc.n[0] = n[0]; pc.n[1] = n[1]; pc.n[2] = n[2];
        pc.d = -(n[0] * X[i1] + n[1] * Y[i1] + n[2] * Z[i1]);
    }
    cand[r] = pc;
}

// Every candidate plane scored in ONE pass over the points.  A wave owns a PATCH of the grid, 64 columns x PTS rows (lane =
// column, so every row is one coalesced load), keeps its points in registers and walks the plane list held in LDS.
//
// The reference's test is fabs(((a*x + b*y) + c*z) + d) < thr in fp64 for every point and plane (PovMesh.cpp:717-742).  Two
// shortcuts, both exact:
//  1. PATCH BOUNDS.  A patch is a compact piece of a smooth surface.  With its bounding box (centre c, half extents e) every
//     point p of the patch has |n.p + d - (n.c + d)| <= |a| ex + |b| ey + |c| ez =: r.  If |n.c + d| + r < thr the whole patch
//     lies inside the band: the plane gets the patch's valid-point count and no point is looked at; if |n.c + d| - r >= thr
//     none does.  Both sides carry a margin (2^-40 of the magnitudes involved) that covers the rounding of the bound itself
//     and of the reference's own fp64 evaluation.  For a bad candidate the band crosses the image as a strip and almost every
//     patch is outside it; for a good one almost every patch is inside.
//  2. For the patches that straddle a band edge the decision per point is taken in PACKED FP32 (v_pk_fma_f32, two points per
//     instruction) against thr -+ margin, margin = 2^-21 ((|a|+|b|+|c|) M + |d| + thr) with M = the largest |coordinate| of the
//     thread's points, which bounds the fp32-fp64 difference (inputs rounded to fp32, three fused roundings:
//     <= 5 * 2^-24 * (|a x| + |b y| + |c z| + |d|)); a wave that meets a point between the two thresholds recounts that plane
//     with the reference's fp64 expression.
// The counts are therefore EXACTLY the reference's (tests compare 400 planes x 4.8 M points).
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int PTS>
__global__ void __launch_bounds__(256) k_ransac_score(const uint8_t* __restrict__ valid, const double* __restrict__ X,
                                                      const double* __restrict__ Y, const double* __restrict__ Z, int w, int h,
                                                      const PlaneCand* __restrict__ cand, int rounds, double thr,
                                                      unsigned long long* __restrict__ counts)
{
    static_assert(PTS % 2 == 0, "points are processed in packed pairs");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double* pl = (double*)smem;                                   // [rounds][4]
    unsigned int* lc = (unsigned int*)(pl + (size_t)rounds * 4);  // [rounds]
    for (int r = threadIdx.x; r < rounds; r += 256) {
        pl[r * 4] = cand[r].n[0]; pl[r * 4 + 1] = cand[r].n[1]; pl[r * 4 + 2] = cand[r].n[2]; pl[r * 4 + 3] = cand[r].d;
        lc[r] = 0;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    // patch of this wave: columns [64 (4 bx + wv), +64), rows [PTS by, +PTS)
    const int col = (blockIdx.x * 4 + wv) * 64 + lane, row0 = blockIdx.y * PTS;
    f32x2 px[PTS / 2], py[PTS / 2], pz[PTS / 2];
    unsigned long long vmask[PTS];
    double mnx = 1e300, mny = 1e300, mnz = 1e300, mxx = -1e300, mxy = -1e300, mxz = -1e300, m = 0.0;
    unsigned int nvalid = 0;
#pragma unroll
    for (int k = 0; k < PTS; ++k) {
        const int row = row0 + k;
        const bool pv = col < w && row < h && valid[(size_t)row * w + col];
        const size_t i = (size_t)row * w + col;
        const double x = pv ? X[i] : 0.0, y = pv ? Y[i] : 0.0, z = pv ? Z[i] : 0.0;
        vmask[k] = __builtin_amdgcn_ballot_w64(pv);
        nvalid += (unsigned)__popcll(vmask[k]);
        if (pv) {
            mnx = fmin(mnx, x); mxx = fmax(mxx, x); mny = fmin(mny, y); mxy = fmax(mxy, y); mnz = fmin(mnz, z); mxz = fmax(mxz, z);
            m = fmax(m, fmax(fabs(x), fmax(fabs(y), fabs(z))));
        }
        px[k / 2][k % 2] = (float)x; py[k / 2][k % 2] = (float)y; pz[k / 2][k % 2] = (float)z;
    }
    if (nvalid) {                                                 // wave-uniform
        for (int o = 32; o > 0; o >>= 1) {
            mnx = fmin(mnx, __shfl_xor(mnx, o)); mxx = fmax(mxx, __shfl_xor(mxx, o));
            mny = fmin(mny, __shfl_xor(mny, o)); mxy = fmax(mxy, __shfl_xor(mxy, o));
            mnz = fmin(mnz, __shfl_xor(mnz, o)); mxz = fmax(mxz, __shfl_xor(mxz, o));
        }
        const double cx = 0.5 * (mnx + mxx), cy = 0.5 * (mny + mxy), cz = 0.5 * (mnz + mxz);
        // half extents, widened so that c +- e really contains min and max after the roundings above
        const double ex = (0.5 * (mxx - mnx)) * (1.0 + 0x1p-50) + 0x1p-1000, ey = (0.5 * (mxy - mny)) * (1.0 + 0x1p-50) + 0x1p-1000,
                     ez = (0.5 * (mxz - mnz)) * (1.0 + 0x1p-50) + 0x1p-1000;
        const float up = 1.0f + 0x1p-20f, G = 0x1p-21f;
        const float Mg = (float)m * up * G, thr_f = (float)thr, thr_g = (float)fabs(thr) * up * G;
        // 64 planes at a time: lane l takes the patch test for plane r0 + l (so the test costs the wave ~2 cycles per plane),
        // the planes it cannot decide are then scored point by point, one after the other
        for (int r0 = 0; r0 < rounds; r0 += 64) {
            const int rl = r0 + lane;
            const bool act = rl < rounds;
            int dec = 2;
            if (act) {
                const double a = pl[rl * 4], b = pl[rl * 4 + 1], c = pl[rl * 4 + 2], d = pl[rl * 4 + 3];
                const double tc = fabs(a * cx + b * cy + c * cz + d);
                const double rad = fabs(a) * ex + fabs(b) * ey + fabs(c) * ez;
                const double slack = 0x1p-40 * (fabs(a * cx) + fabs(b * cy) + fabs(c * cz) + fabs(d) + rad + fabs(thr));
                dec = (tc + rad + slack < thr) ? 1 : ((tc - rad - slack >= thr) ? 2 : 0);
            }
            unsigned int mine = dec == 1 ? nvalid : 0u;           // lane l holds the count of round r0 + l
            unsigned long long und = __builtin_amdgcn_ballot_w64(dec == 0);
            while (und) {                                         // wave-uniform
                const int j = __builtin_ctzll(und);
                und &= und - 1;
                const int r = r0 + j;
                const double a = pl[r * 4], b = pl[r * 4 + 1], c = pl[r * 4 + 2], d = pl[r * 4 + 3];
                const float af = (float)a, bf = (float)b, cf = (float)c, df = (float)d;
                const float nr = (float)(fabs(a) + fabs(b) + fabs(c)) * up;
                const float margin = __builtin_fmaf(nr, Mg, (float)fabs(d) * up * G + thr_g);
                const float lo = thr_f - margin, hi = thr_f + margin;
                const f32x2 a2 = { af, af }, b2 = { bf, bf }, c2 = { cf, cf }, d2 = { df, df };
                unsigned int cnt = 0, cnt_maybe = 0;
#pragma unroll
                for (int k = 0; k < PTS / 2; ++k) {
                    const f32x2 t = __builtin_elementwise_fma(a2, px[k], __builtin_elementwise_fma(b2, py[k], __builtin_elementwise_fma(c2, pz[k], d2)));
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const float at = __builtin_fabsf(t[e]);
                        const unsigned long long in = __builtin_amdgcn_ballot_w64(at < lo) & vmask[2 * k + e];
                        const unsigned long long maybe = __builtin_amdgcn_ballot_w64(!(at >= hi)) & vmask[2 * k + e];     // also true for NaN
                        cnt += (unsigned)__popcll(in);
                        cnt_maybe += (unsigned)__popcll(maybe);    // in is a subset of maybe: equal counts <=> no point in between
                    }
                }
                if (cnt != cnt_maybe) {                           // wave-uniform and rare: this plane again, the reference's way
                    cnt = 0;
                    for (int k = 0; k < PTS; ++k) {
                        const int row = row0 + k;
                        const bool pv = col < w && row < h && valid[(size_t)row * w + col];
                        const size_t i = (size_t)row * w + col;
                        const double x = pv ? X[i] : 0.0, y = pv ? Y[i] : 0.0, z = pv ? Z[i] : 0.0;
                        cnt += (unsigned)__popcll(__builtin_amdgcn_ballot_w64(pv && fabs((a * x + b * y + c * z) + d) < thr));
                    }
                }
                mine += lane == j ? cnt : 0u;
            }
            if (act && mine) atomicAdd(&lc[rl], mine);
        }
    }
    __syncthreads();
    for (int r = threadIdx.x; r < rounds; r += 256)
        if (lc[r]) atomicAdd(&counts[r], (unsigned long long)lc[r]);
}

// ------------------------------------------------------------------ crop_plane (PovMesh.cpp:780-815)
__global__ void __launch_bounds__(256) k_crop_plane(uint8_t* __restrict__ valid, const double* __restrict__ X,
                                                    const double* __restrict__ Y, const double* __restrict__ Z, size_t n,
                                                    double a, double b, double c, double d, double thr,
                                                    unsigned long long* __restrict__ kept)
{
    unsigned int cnt = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        if (valid[i]) {
            if (fabs((a * X[i] + b * Y[i] + c * Z[i]) + d) < thr) ++cnt; else valid[i] = 0;
        }
    }
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_down(cnt, o);
    if ((threadIdx.x & 63) == 0 && cnt) atomicAdd(kept + slot_of_block(), (unsigned long long)cnt);
}

// first strictly better candidate wins (PovMesh.cpp:750-755); failure if best < W*H/10 (:773)
__global__ void __launch_bounds__(64) k_ransac_pick(const PlaneCand* __restrict__ cand, const unsigned long long* __restrict__ counts,
                                                    int rounds, size_t n, DevState* __restrict__ ds)
{
    // key = (count << 32) | ~round : the largest count wins, the earliest round among equals (strict '>' in :750)
    unsigned long long key = 0;
    for (int r = threadIdx.x; r < rounds; r += 64)
        if (cand[r].ok && counts[r] > 0) {
            const unsigned long long k = (counts[r] << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)r);
            key = k > key ? k : key;
        }
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long other = __shfl_down(key, o);
        key = other > key ? other : key;
    }
    if (threadIdx.x == 0) {
        const unsigned long long best = key >> 32;
        double pl[4] = { 0, 0, 0, 0 };
        if (best) {
            const int r = (int)(0xFFFFFFFFu - (unsigned)(key & 0xFFFFFFFFull));
            pl[0] = cand[r].n[0]; pl[1] = cand[r].n[1]; pl[2] = cand[r].n[2]; pl[3] = cand[r].d;
        }
        for (int k = 0; k < 4; ++k) ds->ransac_plane[k] = pl[k];
        ds->ransac_best = best;
        ds->ransac_found = best < n / 10 ? 0 : 1;
    }
}
// crop_plane with the plane (and the "RANSAC succeeded" switch) read from device memory
__global__ void __launch_bounds__(256) k_crop_plane_dev(uint8_t* __restrict__ valid, const double* __restrict__ X,
                                                        const double* __restrict__ Y, const double* __restrict__ Z, size_t n,
                                                        const double* __restrict__ plane, const int* __restrict__ enable,
                                                        double thr, unsigned long long* __restrict__ kept)
{
    if (!*enable) return;
    const double a = plane[0], b = plane[1], c = plane[2], d = plane[3];
    unsigned int cnt = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        if (valid[i]) {
            if (fabs((a * X[i] + b * Y[i] + c * Z[i]) + d) < thr) ++cnt; else valid[i] = 0;
        }
    }
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_down(cnt, o);
    if ((threadIdx.x & 63) == 0 && cnt) atomicAdd(kept + slot_of_block(), (unsigned long long)cnt);
}

// ------------------------------------------------------------------ refine_plane (PovMesh.cpp:581-660)
struct RefineDev { double xmin, xmax, ymin, ymax, maxd; int weighted, umin, umax, vmin, vmax; };

// block-level sum of NV doubles; partial sums go to out[block][NV] and are added on the host in block
// order, so the result does not depend on scheduling
template <int NV>
__device__ __forceinline__ void block_sum_store(double (&v)[NV], double* __restrict__ out)
{
    __shared__ double sh[4][NV];
#pragma unroll
    for (int k = 0; k < NV; ++k)
        for (int o = 32; o > 0; o >>= 1) v[k] += __shfl_down(v[k], o);
    const int wv = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0)
#pragma unroll
        for (int k = 0; k < NV; ++k) sh[wv][k] = v[k];
    __syncthreads();
    if (threadIdx.x == 0)
#pragma unroll
        for (int k = 0; k < NV; ++k) out[(size_t)blockIdx.x * NV + k] = ((sh[0][k] + sh[1][k]) + sh[2][k]) + sh[3][k];
}

__device__ __forceinline__ bool refine_inlier(const RefineDev& rp, const uint8_t* valid, const double* X, const double* Y,
                                              const double* Z, int w, size_t i, double& px, double& py, double& pz, double& wt)
{
    const unsigned int ii = (unsigned int)i;                     // a mesh has fewer than 2^31 points: 32-bit division
    const int u = (int)(ii % (unsigned int)w), v = (int)(ii / (unsigned int)w);
    if (u < rp.umin || u > rp.umax || v < rp.vmin || v > rp.vmax || !valid[i]) return false;
    px = X[i]; py = Y[i]; pz = Z[i];
    const double dist = sqrt(px * px + py * py + pz * pz);
    if (!(px > rp.xmin && px < rp.xmax && py > rp.ymin && py < rp.ymax && dist < rp.maxd)) return false;
    wt = rp.weighted ? dist : 1.0;
    return true;
}

// pass 0: {count, wsum, sum w*x, sum w*y, sum w*z}
__global__ void __launch_bounds__(256) k_refine_moments(const uint8_t* __restrict__ valid, const double* __restrict__ X,
                                                        const double* __restrict__ Y, const double* __restrict__ Z, int w,
                                                        size_t n, RefineDev rp, double* __restrict__ partial)
{
    double acc[5] = { 0, 0, 0, 0, 0 };
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        double px, py, pz, wt;
        if (refine_inlier(rp, valid, X, Y, Z, w, i, px, py, pz, wt)) {
            acc[0] += 1.0; acc[1] += wt; acc[2] += px * wt; acc[3] += py * wt; acc[4] += pz * wt;
        }
    }
    block_sum_store<5>(acc, partial);
}
// crop_plane by the RANSAC plane (k_crop_plane_dev) and pass 0 of the refinement in ONE pass over the points: the same
// walk as k_refine_moments (grid, stride and therefore summation order), with the crop decision taken -- and written to
// valid -- just before the point is offered to the moments.
__global__ void __launch_bounds__(256) k_crop_moments_dev(uint8_t* __restrict__ valid, const double* __restrict__ X,
                                                          const double* __restrict__ Y, const double* __restrict__ Z, int w, size_t n,
                                                          const double* __restrict__ plane, const int* __restrict__ enable, double thr,
                                                          unsigned long long* __restrict__ kept, RefineDev rp,
                                                          double* __restrict__ partial)
{
    const bool crop = *enable != 0;
    const double a = plane[0], b = plane[1], c = plane[2], d = plane[3];
    unsigned int cnt = 0;
    double acc[5] = { 0, 0, 0, 0, 0 };
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        if (!valid[i]) continue;
        const double px = X[i], py = Y[i], pz = Z[i];
        if (crop) {
            if (fabs((a * px + b * py + c * pz) + d) < thr) ++cnt;
            else { valid[i] = 0; continue; }
        }
        const unsigned int ii = (unsigned int)i;
        const int u = (int)(ii % (unsigned int)w), v = (int)(ii / (unsigned int)w);
        if (u < rp.umin || u > rp.umax || v < rp.vmin || v > rp.vmax) continue;
        const double dist = sqrt(px * px + py * py + pz * pz);
        if (!(px > rp.xmin && px < rp.xmax && py > rp.ymin && py < rp.ymax && dist < rp.maxd)) continue;
        const double wt = rp.weighted ? dist : 1.0;
        acc[0] += 1.0; acc[1] += wt; acc[2] += px * wt; acc[3] += py * wt; acc[4] += pz * wt;
    }
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_down(cnt, o);
    if ((threadIdx.x & 63) == 0 && cnt) atomicAdd(kept + slot_of_block(), (unsigned long long)cnt);
    block_sum_store<5>(acc, partial);
}
// pass 1: weighted scatter matrix around the centroid (6 unique entries)
__global__ void __launch_bounds__(256) k_refine_cov(const uint8_t* __restrict__ valid, const double* __restrict__ X,
                                                    const double* __restrict__ Y, const double* __restrict__ Z, int w, size_t n,
                                                    RefineDev rp, double cx, double cy, double cz,
                                                    double* __restrict__ partial)
{
    double acc[6] = { 0, 0, 0, 0, 0, 0 };
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        double px, py, pz, wt;
        if (refine_inlier(rp, valid, X, Y, Z, w, i, px, py, pz, wt)) {
            const double qx = px - cx, qy = py - cy, qz = pz - cz;
            acc[0] += wt * qx * qx; acc[1] += wt * qx * qy; acc[2] += wt * qx * qz;
            acc[3] += wt * qy * qy; acc[4] += wt * qy * qz; acc[5] += wt * qz * qz;
        }
    }
    block_sum_store<6>(acc, partial);
}

// plane_refinement_inliers.xyz (wass_stereo.cpp:2077-2085 writes every 10th refinement inlier, in raster order): the
// selection runs on the device so that the host fetches ~150 000 points instead of the whole 126 MB mesh.
// pass 1: refinement inliers per block of 256 points; pass 2 (after the scan): inlier number k of the raster order is
// kept when k % every == 0, at position k / every.
__global__ void __launch_bounds__(256) k_inlier_counts(const uint8_t* __restrict__ valid, const double* __restrict__ X,
                                                       const double* __restrict__ Y, const double* __restrict__ Z, int w, size_t n,
                                                       RefineDev rp, unsigned int* __restrict__ blockcnt)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    double px, py, pz, wt;
    const bool in = i < n && refine_inlier(rp, valid, X, Y, Z, w, i, px, py, pz, wt);
    const int c = __syncthreads_count(in);
    if (threadIdx.x == 0) blockcnt[blockIdx.x] = (unsigned)c;
}
__global__ void __launch_bounds__(256) k_inlier_pack(const uint8_t* __restrict__ valid, const double* __restrict__ X,
                                                     const double* __restrict__ Y, const double* __restrict__ Z, int w, size_t n,
                                                     RefineDev rp, const unsigned int* __restrict__ blockoff, unsigned int every,
                                                     double* __restrict__ out)
{
    __shared__ unsigned int wsum[4];
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    double px = 0, py = 0, pz = 0, wt;
    const bool in = i < n && refine_inlier(rp, valid, X, Y, Z, w, i, px, py, pz, wt);
    const unsigned long long bal = __ballot(in);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) wsum[wv] = (unsigned)__popcll(bal);
    __syncthreads();
    unsigned int k = blockoff[blockIdx.x];
    for (int q = 0; q < wv; ++q) k += wsum[q];
    k += (unsigned)__popcll(bal & ((1ull << lane) - 1ull));
    if (in && k % every == 0) {
        double* o = out + (size_t)(k / every) * 3;
        o[0] = px; o[1] = py; o[2] = pz;
    }
}

// ---- plane_refinement_inliers.xyz as TEXT, on the device (round 5).  The reference writes "x y z\n" per selected inlier through a
// default ofstream (wass_stereo.cpp:2077-2085): six significant digits, %g.  fmt_g6.h produces exactly those characters; a line
// is at most 3 x 12 + 3 = 39 bytes.  Two passes over the selected points (their number is only known on the device): line
// lengths per block of 256 points -> k_scan_blocks -> the lines written at their offsets.  A number outside the formatter's
// domain (inf, nan, |v| >= 1e6 or < 1e-22: never a triangulated coordinate) is counted in *bad and the host formats the file
// from the doubles instead, as before.
constexpr int INL_LINE_MAX = 40;
// pass 1: a block formats 256 lines into LDS (a thread's line is a string of single bytes: in HBM that is 40 partial cache-line
// writes per thread, 64 different lines per instruction), packs them back to back -- still in LDS -- and writes the packed chunk to
// the block's staging area as dwords; blockbytes[b] = its length.  pass 2 (after the scan): the chunks copied to their final
// offsets, again as dwords (the destination's alignment differs from the source's: two source dwords and a byte funnel shift per
// destination dword).  Byte-granular global traffic made these two kernels 0.36 ms per frame, which a saturated GPU charges in full.
__global__ void __launch_bounds__(256) k_inl_text_format(const double* __restrict__ pts, const unsigned int* __restrict__ total, unsigned int every,
                                                         char* __restrict__ staging, unsigned int* __restrict__ blockbytes, unsigned int* __restrict__ bad_out)
{
    __shared__ __attribute__((aligned(16))) char lines[256][INL_LINE_MAX];
    __shared__ __attribute__((aligned(16))) char packed[256 * INL_LINE_MAX + 16];
    __shared__ unsigned int wsum[4];
    const unsigned int nsel = (*total + every - 1) / every;
    const unsigned int j = blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    unsigned int len = 0, bad = 0;
    if (j < nsel) {
        char* const buf = lines[threadIdx.x];
        const double* p = pts + (size_t)j * 3;
        int n = 0;
#pragma unroll 1
        for (int k = 0; k < 3; ++k) {
            int l = fmt_g6(p[k], buf + n);
            if (l < 0) { ++bad; buf[n] = '?'; l = 1; }
            n += l;
            buf[n++] = k < 2 ? ' ' : '\n';
        }
        len = (unsigned)n;
    }
    unsigned int incl = len;                               // inclusive scan over the wave, then over the block's four waves
    for (int o = 1; o < 64; o <<= 1) { const unsigned int t = __shfl_up(incl, o); if (lane >= o) incl += t; }
    if (lane == 63) wsum[wv] = incl;
    __syncthreads();
    unsigned int off = incl - len;
    for (int q = 0; q < wv; ++q) off += wsum[q];
    const unsigned int btot = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    for (unsigned int i = 0; i < len; ++i) packed[off + i] = lines[threadIdx.x][i];     // LDS to LDS: every thread its own line
    __syncthreads();
    uint32_t* const dst = (uint32_t*)(staging + (size_t)blockIdx.x * (256 * INL_LINE_MAX));
    const uint32_t* const src = (const uint32_t*)packed;
    for (unsigned int i = threadIdx.x; i < (btot + 3) / 4; i += 256) dst[i] = src[i];   // (the last dword's pad bytes are never copied on)
    if (threadIdx.x == 0) blockbytes[blockIdx.x] = btot;
    if (bad) atomicAdd(bad_out, bad);
}
__global__ void __launch_bounds__(256) k_inl_text_pack(const char* __restrict__ staging, const unsigned int* __restrict__ blockoff,
                                                       const unsigned int* __restrict__ text_total, unsigned int nblocks, char* __restrict__ out)
{
    const unsigned int b = blockIdx.x;
    const unsigned int o0 = blockoff[b], o1 = b + 1 < nblocks ? blockoff[b + 1] : *text_total;
    const unsigned int nbytes = o1 - o0;
    if (nbytes == 0) return;
    const char* src = staging + (size_t)b * (256 * INL_LINE_MAX);                       // dword aligned
    char* const d = out + o0;
    // head bytes up to the destination's next dword boundary, whole dwords, tail bytes
    const unsigned int head = min(nbytes, (unsigned int)((4 - ((uintptr_t)d & 3)) & 3));
    if (threadIdx.x < head) d[threadIdx.x] = src[threadIdx.x];
    const unsigned int nd = (nbytes - head) / 4;
    const uint32_t* const s32 = (const uint32_t*)src;
    uint32_t* const d32 = (uint32_t*)(d + head);
    for (unsigned int i = threadIdx.x; i < nd; i += 256) {
        const uint32_t lo = s32[i], hi = head ? s32[i + 1] : 0u;
        d32[i] = head ? __builtin_amdgcn_alignbyte(hi, lo, head) : lo;
    }
    const unsigned int done = head + 4 * nd;
    if (threadIdx.x < nbytes - done) d[done + threadIdx.x] = src[done + threadIdx.x];
}

__global__ void k_inl_text_totals(DevState* __restrict__ ds, const unsigned int* __restrict__ text_total)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        const bool have = ds->ransac_found != 0;
        ds->inl_text_bytes = have ? text_total[0] : 0ull;
        ds->inl_text_bad = have ? text_total[1] : 0u;
    }
}

// smallest-eigenvalue eigenvector of a symmetric 3x3 (cyclic Jacobi); stands in for row 2 of cv::SVD's vt
__host__ __device__ static void smallest_eigvec3(const double Ain[9], double vout[3])
{
    double A[3][3], V[3][3] = { { 1, 0, 0 }, { 0, 1, 0 }, { 0, 0, 1 } };
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) A[i][j] = Ain[i * 3 + j];
    for (int it = 0; it < 64; it++) {
        const double off = fabs(A[0][1]) + fabs(A[0][2]) + fabs(A[1][2]);
        const double diag = fabs(A[0][0]) + fabs(A[1][1]) + fabs(A[2][2]);
        if (off <= 1e-300 || off <= 1e-22 * diag) break;
        for (int i = 0; i < 2; i++)
            for (int j = i + 1; j < 3; j++) {
                if (A[i][j] == 0.0) continue;
                const double theta = (A[j][j] - A[i][i]) / (2.0 * A[i][j]);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double cs = 1.0 / sqrt(t * t + 1.0), sn = t * cs;
                for (int k = 0; k < 3; k++) { const double a = A[i][k], b = A[j][k]; A[i][k] = cs * a - sn * b; A[j][k] = sn * a + cs * b; }
                for (int k = 0; k < 3; k++) { const double a = A[k][i], b = A[k][j]; A[k][i] = cs * a - sn * b; A[k][j] = sn * a + cs * b; }
                for (int k = 0; k < 3; k++) { const double a = V[k][i], b = V[k][j]; V[k][i] = cs * a - sn * b; V[k][j] = sn * a + cs * b; }
            }
    }
    int m = 0;
    for (int i = 1; i < 3; i++) if (A[i][i] < A[m][m]) m = i;
    const double nn = sqrt(V[0][m] * V[0][m] + V[1][m] * V[1][m] + V[2][m] * V[2][m]);
    vout[0] = V[0][m] / nn; vout[1] = V[1][m] / nn; vout[2] = V[2][m] / nn;
}


// Sum of per-block partials in a fixed two-level order (64 strided sub-sums, then those in index order): the same
// function runs on the host (step-by-step API) and on the device (fused API), so both give identical bits.
template <int NV>
__host__ __device__ static void sum_partials_lane(const double* partial, int nb, int lane, double (&out)[NV])
{
    for (int k = 0; k < NV; ++k) out[k] = 0;
    for (int b = lane; b < nb; b += 64)
        for (int k = 0; k < NV; ++k) out[k] += partial[(size_t)b * NV + k];
}
template <int NV>
__device__ static void sum_partials_wave(const double* __restrict__ partial, int nb, double (&out)[NV])
{
    double mine[NV];
    sum_partials_lane<NV>(partial, nb, threadIdx.x & 63, mine);
    for (int k = 0; k < NV; ++k) {
        double tot = 0;
        for (int l = 0; l < 64; ++l) tot += __shfl(mine[k], l);          // lane order, every lane computes the same total
        out[k] = tot;
    }
}
template <int NV>
static void sum_partials_host(const double* partial, int nb, double (&out)[NV])
{
    double lanes[64][NV];
    for (int l = 0; l < 64; ++l) sum_partials_lane<NV>(partial, nb, l, lanes[l]);
    for (int k = 0; k < NV; ++k) { double tot = 0; for (int l = 0; l < 64; ++l) tot += lanes[l][k]; out[k] = tot; }
}

// partial sums of k_refine_moments -> centroid
__global__ void k_refine_centroid(const double* __restrict__ partial, int nb, const int* __restrict__ enable, DevState* __restrict__ ds)
{
    if (blockIdx.x) return;
    if (!*enable) { if (threadIdx.x == 0) ds->refine_ok = 0; return; }
    double mom[5];
    sum_partials_wave<5>(partial, nb, mom);
    if (threadIdx.x) return;
    ds->ninl = mom[0]; ds->wsum = mom[1];
    ds->refine_ok = (mom[0] >= 3 && mom[1] > 0) ? 1 : 0;
    ds->centroid[0] = mom[2] / mom[1]; ds->centroid[1] = mom[3] / mom[1]; ds->centroid[2] = mom[4] / mom[1];
}
__global__ void __launch_bounds__(256) k_refine_cov_dev(const uint8_t* __restrict__ valid, const double* __restrict__ X,
                                                        const double* __restrict__ Y, const double* __restrict__ Z, int w, size_t n,
                                                        RefineDev rp, const DevState* __restrict__ ds, double* __restrict__ partial)
{
    const double cx = ds->centroid[0], cy = ds->centroid[1], cz = ds->centroid[2];
    const bool on = ds->refine_ok != 0;
    double acc[6] = { 0, 0, 0, 0, 0, 0 };
    if (on)
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
            double px, py, pz, wt;
            if (refine_inlier(rp, valid, X, Y, Z, w, i, px, py, pz, wt)) {
                const double qx = px - cx, qy = py - cy, qz = pz - cz;
                acc[0] += wt * qx * qx; acc[1] += wt * qx * qy; acc[2] += wt * qx * qz;
                acc[3] += wt * qy * qy; acc[4] += wt * qy * qz; acc[5] += wt * qz * qz;
            }
        }
    block_sum_store<6>(acc, partial);
}
// scatter matrix -> plane (PovMesh.cpp:641-656)
__global__ void k_refine_finish(const double* __restrict__ partial, int nb, DevState* __restrict__ ds)
{
    if (blockIdx.x || !ds->refine_ok) return;
    double s[6];
    sum_partials_wave<6>(partial, nb, s);
    if (threadIdx.x) return;
    const double A[9] = { s[0], s[1], s[2], s[1], s[3], s[4], s[2], s[4], s[5] };
    double nrm[3];
    smallest_eigvec3(A, nrm);
    const double nn = sqrt(nrm[0] * nrm[0] + nrm[1] * nrm[1] + nrm[2] * nrm[2]);
    nrm[0] /= nn; nrm[1] /= nn; nrm[2] /= nn;
    if (nrm[2] < 0) { nrm[0] *= -1.0; nrm[1] *= -1.0; nrm[2] *= -1.0; }
    ds->plane[0] = nrm[0]; ds->plane[1] = nrm[1]; ds->plane[2] = nrm[2];
    ds->plane[3] = -(nrm[0] * ds->centroid[0] + nrm[1] * ds->centroid[1] + nrm[2] * ds->centroid[2]);
}

// ------------------------------------------------------------------ xyzC (PovMesh.cpp:377-460)
struct RTDev { double R[9], T[3]; };

__device__ __forceinline__ unsigned long long dkey(double v)     // order-preserving map double -> u64
{
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    return (b & 0x8000000000000000ull) ? ~b : (b | 0x8000000000000000ull);
}
__host__ __device__ __forceinline__ double dunkey(unsigned long long k)
{
    const unsigned long long b = (k & 0x8000000000000000ull) ? (k & 0x7FFFFFFFFFFFFFFFull) : ~k;
    double d;
    memcpy(&d, &b, 8);
    return d;
}
// min/max of R*p+T per axis (exact, order independent) + number of valid points per block
__global__ void __launch_bounds__(256) k_block_counts(const uint8_t* __restrict__ valid, size_t n,
                                                      unsigned int* __restrict__ blockcnt)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const int c = __syncthreads_count(i < n && valid[i]);
    if (threadIdx.x == 0) blockcnt[blockIdx.x] = (unsigned)c;
}
__device__ __forceinline__ void xyzc_limits_body(const uint8_t* __restrict__ valid, const double* __restrict__ X,
                                                 const double* __restrict__ Y, const double* __restrict__ Z, size_t n,
                                                 const RTDev& rt, unsigned long long* __restrict__ lim /* [NSLOT][6] keys */)
{
    unsigned long long mn[3] = { ~0ull, ~0ull, ~0ull }, mx[3] = { 0, 0, 0 };
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        if (!valid[i]) continue;
        const double p[3] = { X[i], Y[i], Z[i] };
        double t[3];
        mulv(rt.R, p, t);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const unsigned long long key = dkey(t[k] + rt.T[k]);
            mn[k] = key < mn[k] ? key : mn[k];
            mx[k] = key > mx[k] ? key : mx[k];
        }
    }
#pragma unroll
    for (int k = 0; k < 3; ++k)
        for (int o = 32; o > 0; o >>= 1) {
            const unsigned long long a = __shfl_down(mn[k], o), b = __shfl_down(mx[k], o);
            mn[k] = a < mn[k] ? a : mn[k];
            mx[k] = b > mx[k] ? b : mx[k];
        }
    if ((threadIdx.x & 63) == 0) {
        unsigned long long* l = lim + (size_t)slot_of_block() * 6;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            if (mn[k] != ~0ull) atomicMin(&l[k], mn[k]);
            if (mx[k] != 0) atomicMax(&l[3 + k], mx[k]);
        }
    }
}
__global__ void __launch_bounds__(256) k_xyzc_limits(const uint8_t* __restrict__ valid, const double* __restrict__ X,
                                                     const double* __restrict__ Y, const double* __restrict__ Z, size_t n,
                                                     RTDev rt, unsigned long long* __restrict__ lim)
{
    xyzc_limits_body(valid, X, Y, Z, n, rt, lim);
}
// exclusive scan of the per-block counts (single block; nblocks <= ~25k at full size)
__global__ void __launch_bounds__(1024) k_scan_blocks(unsigned int* __restrict__ cnt, int nb, unsigned int* __restrict__ total)
{
    // exclusive scan of nb counts by one workgroup: each of the 16 waves owns a contiguous range and reads it 64 at a time
    // (coalesced, independent loads); range totals meet in LDS, then every wave rewrites its range with a running carry
    // any number of waves up to 16 (blockDim.x / 64).  Launched with 256 threads in the frame tail: a 1024-thread workgroup needs four free
    // wave slots on every SIMD of ONE compute unit, and underneath the next frame's aggregation kernels (five waves per SIMD) that can
    // mean waiting for one of their workgroups to retire -- with the whole tail stream queued behind it.
    __shared__ unsigned int wtot[16];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = (int)(blockDim.x >> 6);
    const int per = ((nb + nw - 1) / nw + 63) & ~63, begin = min(nb, wv * per), end = min(nb, begin + per);
    unsigned int sum = 0;
    for (int i = begin + lane; i < end; i += 64) sum += cnt[i];
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_down(sum, o);
    if (lane == 0) wtot[wv] = sum;
    __syncthreads();
    unsigned int carry = 0, all = 0;
    for (int k = 0; k < nw; ++k) { if (k < wv) carry += wtot[k]; all += wtot[k]; }
    if (threadIdx.x == 0) *total = all;
    for (int base = begin; base < end; base += 64) {
        const int i = base + lane;
        const unsigned int v = i < end ? cnt[i] : 0;
        unsigned int incl = v;
        for (int o = 1; o < 64; o <<= 1) { const unsigned int t = __shfl_up(incl, o); if (lane >= o) incl += t; }
        if (i < end) cnt[i] = carry + incl - v;
        carry += __shfl(incl, 63);
    }
}
__device__ __forceinline__ void xyzc_pack_body(const uint8_t* __restrict__ valid, const double* __restrict__ X,
                                               const double* __restrict__ Y, const double* __restrict__ Z, size_t n,
                                               const RTDev& rt, double mnx, double mny, double mnz, double sx, double sy,
                                               double sz, const unsigned int* __restrict__ blockoff, uint16_t* __restrict__ out)
{
    __shared__ unsigned int wsum[4];
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const bool v = i < n && valid[i];
    const unsigned long long bal = __ballot(v);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) wsum[wv] = (unsigned)__popcll(bal);
    __syncthreads();
    unsigned int off = blockoff[blockIdx.x];
    for (int k = 0; k < wv; ++k) off += wsum[k];
    off += (unsigned)__popcll(bal & ((1ull << lane) - 1ull));
    if (v) {
        const double p[3] = { X[i], Y[i], Z[i] };
        double t[3];
        mulv(rt.R, p, t);
        t[0] += rt.T[0]; t[1] += rt.T[1]; t[2] += rt.T[2];
        out[(size_t)off * 3 + 0] = (uint16_t)((t[0] - mnx) * sx);
        out[(size_t)off * 3 + 1] = (uint16_t)((t[1] - mny) * sy);
        out[(size_t)off * 3 + 2] = (uint16_t)((t[2] - mnz) * sz);
    }
}
__global__ void __launch_bounds__(256) k_xyzc_pack(const uint8_t* __restrict__ valid, const double* __restrict__ X,
                                                   const double* __restrict__ Y, const double* __restrict__ Z, size_t n,
                                                   RTDev rt, double mnx, double mny, double mnz, double sx, double sy,
                                                   double sz, const unsigned int* __restrict__ blockoff,
                                                   uint16_t* __restrict__ out)
{
    xyzc_pack_body(valid, X, Y, Z, n, rt, mnx, mny, mnz, sx, sy, sz, blockoff, out);
}

__global__ void __launch_bounds__(256) k_interleave(const double* __restrict__ X, const double* __restrict__ Y,
                                                    const double* __restrict__ Z, size_t n, double* __restrict__ out)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) { out[3 * i] = X[i]; out[3 * i + 1] = Y[i]; out[3 * i + 2] = Z[i]; }
}
__global__ void __launch_bounds__(256) k_deinterleave(const double* __restrict__ in, size_t n, double* __restrict__ X,
                                                      double* __restrict__ Y, double* __restrict__ Z)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) { X[i] = in[3 * i]; Y[i] = in[3 * i + 1]; Z[i] = in[3 * i + 2]; }
}

// hipMalloc / hipFree cost milliseconds and synchronise the device, so destroyed meshes park their allocation in a
// small pool and the next frame of the same size takes it back.  A mesh is destroyed when its last use has been
// ENQUEUED, not when it has run, so an allocation only ever goes back to the context that owned it: that context's
// stream orders the old kernels before the new ones.  (Another context would start writing while they still run.)
struct PoolEntry { void* p; size_t bytes; int device; const void* owner; };
static PoolEntry g_pool[8];
static std::mutex g_pool_mu;
// contexts that exist: a mesh may be destroyed AFTER its context (a garbage collector picks the order), and an
// allocation parked under a dead owner would never be purged -- or be handed to a new context at the same address
static const void* g_live[64];
void mesh_pool_ctx_alive(const void* ctx, bool alive)
{
    std::lock_guard<std::mutex> lk(g_pool_mu);
    for (auto& e : g_live)
        if (alive ? e == nullptr : e == ctx) { e = alive ? ctx : nullptr; return; }
}
static void* pool_take(size_t bytes, int device, const void* owner)
{
    std::lock_guard<std::mutex> lk(g_pool_mu);
    for (auto& e : g_pool)
        if (e.p && e.bytes == bytes && e.device == device && e.owner == owner) { void* p = e.p; e.p = nullptr; return p; }
    return nullptr;
}
static bool pool_give(void* p, size_t bytes, int device, const void* owner)
{
    std::lock_guard<std::mutex> lk(g_pool_mu);
    bool live = false;
    for (const void* e : g_live) live = live || e == owner;
    if (!live) return false;                                     // the caller frees the allocation
    for (auto& e : g_pool)
        if (!e.p) { e.p = p; e.bytes = bytes; e.device = device; e.owner = owner; return true; }
    return false;
}
// called by wass_ctx_destroy (after the context's streams have drained)
void mesh_pool_purge(const void* owner)
{
    std::lock_guard<std::mutex> lk(g_pool_mu);
    for (auto& e : g_pool)
        if (e.p && e.owner == owner) { (void)hipFree(e.p); e.p = nullptr; }
}

static int mesh_alloc(wass_ctx* c, int w, int h, wass_mesh** out)
{
    wass_mesh* m = new (std::nothrow) wass_mesh();
    if (!m) return set_err(c, WASS_ERR_NO_MEMORY, "out of host memory");
    m->w = w; m->h = h;
    const size_t n = m->n();
    if (n > 0x7FFFFFFFull) { delete m; return set_err(c, WASS_ERR_UNSUPPORTED, "mesh too large (%d x %d)", w, h); }   // kernels index points with 32 bits
    // one allocation: x | y | z | valid | gray | codes
    const size_t bytes = n * 8 * 3 + ((n + 255) & ~(size_t)255) * 3;
    m->bytes = bytes; m->device = c->device; m->owner = c;
    void* base = pool_take(bytes, c->device, c);
    if (!base && hipMalloc(&base, bytes) != hipSuccess) { delete m; return set_err(c, WASS_ERR_NO_MEMORY, "hipMalloc(%zu) failed", bytes); }
    m->x = (double*)base; m->y = m->x + n; m->z = m->y + n;
    m->valid = (uint8_t*)(m->z + n);
    m->gray = m->valid + ((n + 255) & ~(size_t)255);
    m->codes = m->gray + ((n + 255) & ~(size_t)255);
    *out = m;
    return WASS_OK;
}

static inline unsigned nblk(size_t n) { return (unsigned)((n + 255) / 256); }

// striped device counters (NSLOT u64) -> host sum
static int counters_reset(wass_ctx* c, unsigned long long** cnt)
{
    int rc = ensure(c, c->counters, (size_t)NSLOT * 6 * 8);
    if (rc) return rc;
    *cnt = (unsigned long long*)c->counters.p;
    WASS_HIP(c, hipMemsetAsync(*cnt, 0, (size_t)NSLOT * 8, c->ts()));
    return WASS_OK;
}
static int counters_sum(wass_ctx* c, const unsigned long long* cnt, unsigned long long* out)
{
    unsigned long long h[NSLOT];
    WASS_HIP(c, hipMemcpyAsync(h, cnt, sizeof h, hipMemcpyDeviceToHost, c->ts()));
    WASS_HIP(c, hipStreamSynchronize(c->ts()));
    unsigned long long s = 0;
    for (int i = 0; i < NSLOT; ++i) s += h[i];
    *out = s;
    return WASS_OK;
}

}  // namespace wass

using namespace wass;

extern "C" {

void wass_mesh_destroy(wass_mesh* m)
{
    if (!m) return;
    if (m->x && !pool_give(m->x, m->bytes, m->device, m->owner)) { (void)hipSetDevice(m->device); (void)hipFree(m->x); }
    delete m;
}

int wass_mesh_reject_codes(wass_ctx* c, const wass_mesh* m, uint8_t* codes_out)
{
    if (!c || !m || !codes_out) return set_err(c, WASS_ERR_INVALID_ARG, "null argument");
    WASS_HIP(c, hipSetDevice(c->device));
    WASS_HIP(c, hipMemcpyAsync(codes_out, m->codes, m->n(), hipMemcpyDeviceToHost, c->ts()));
    WASS_HIP(c, hipStreamSynchronize(c->ts()));
    return WASS_OK;
}

int wass_mesh_size(const wass_mesh* m, int* width, int* height)
{
    if (!m) return WASS_ERR_INVALID_ARG;
    if (width) *width = m->w;
    if (height) *height = m->h;
    return WASS_OK;
}

void wass_free(void* p) { free(p); }

int wass_triangulate_dev(wass_ctx* c, const float* d_disp, int W, int H, const int roi_l[4], const int roi_r[4],
                         const wass_geom* g, const uint8_t* d_right_img, int img_w, int img_h, const uint8_t* d_lmask,
                         const uint8_t* d_rmask, const wass_tri_params* tp, wass_mesh** out, uint64_t* n_pts)
{
    if (!c || !d_disp || !roi_l || !roi_r || !g || !d_right_img || !tp || !out)
        return set_err(c, WASS_ERR_INVALID_ARG, "null argument");
    if (roi_r[2] <= 0 || roi_r[3] <= 0 || W <= 0 || H <= 0 || img_w <= 0 || img_h <= 0)
        return set_err(c, WASS_ERR_INVALID_ARG, "bad geometry");
    if (g->dense_scale == 0.0) return set_err(c, WASS_ERR_INVALID_ARG, "dense_scale must be non-zero");
    WASS_HIP(c, hipSetDevice(c->device));
    wass_mesh* m = nullptr;
    int rc = mesh_alloc(c, roi_r[2], roi_r[3], &m);
    if (rc) return rc;
    GeomDev gd;
    memcpy(gd.Kl, g->K_left, sizeof gd.Kl); memcpy(gd.Kr, g->K_right, sizeof gd.Kr);
    memcpy(gd.R, g->R, sizeof gd.R); memcpy(gd.T, g->T, sizeof gd.T);
    gd.use_custom = g->use_custom;
    memcpy(gd.R1, g->R1, sizeof gd.R1); memcpy(gd.R2, g->R2, sizeof gd.R2);
    memcpy(gd.P1, g->P1, sizeof gd.P1); memcpy(gd.P2, g->P2, sizeof gd.P2);
    memcpy(gd.HLi, g->HLi, sizeof gd.HLi); memcpy(gd.HRi, g->HRi, sizeof gd.HRi);
    gd.comp_over_scale = g->disparity_compensation / g->dense_scale;
    // counted into a buffer of its own: the frame tail (wass_mesh_finish_frame_async) reports it, long after the shared
    // counters have been reused by the plane stages
    unsigned long long* cnt = nullptr;
    if ((rc = ensure(c, c->tri_cnt, (size_t)NSLOT * 8))) { wass_mesh_destroy(m); return rc; }
    cnt = (unsigned long long*)c->tri_cnt.p;
    if (hipMemsetAsync(cnt, 0, (size_t)NSLOT * 8, c->ts()) != hipSuccess) { wass_mesh_destroy(m); return set_err(c, WASS_ERR_DEVICE, "memset failed"); }
    dim3 grid((m->w + 255) / 256, std::min(m->h, 512));
    if (!c->ev_tail_sets[0][0])
        for (auto& set : c->ev_tail_sets)
            for (auto& e : set)
                if (hipEventCreate(&e) != hipSuccess) { e = nullptr; wass_mesh_destroy(m); return set_err(c, WASS_ERR_DEVICE, "hipEventCreate failed"); }
    c->tail_set = (c->tail_set + 1) & 3;         // a frame's set stays readable for three more triangulations (two frames may be pending)
    c->ev_tail = c->ev_tail_sets[c->tail_set];
    (void)hipEventRecord(c->ev_tail[0], c->ts());
    c->tail_timed[c->tail_set] = false;
    hipLaunchKernelGGL(k_triangulate, grid, dim3(256), 0, c->ts(), d_disp, W, H, roi_l[0], roi_r[0], roi_r[1], m->w, m->h,
                       gd, d_right_img, img_w, img_h, d_lmask, d_rmask, tp->min_angle_deg, tp->bbox[0], tp->bbox[1],
                       tp->bbox[2], tp->bbox[3], tp->cam_distance, m->valid, m->x, m->y, m->z, m->gray, m->codes, cnt);
    unsigned long long hc = 0;
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { wass_mesh_destroy(m); return set_err(c, WASS_ERR_DEVICE, "triangulate: %s", hipGetErrorString(e)); }
    if (n_pts) {                                 // the count is the only reason to synchronise here
        if ((rc = counters_sum(c, cnt, &hc))) { wass_mesh_destroy(m); return rc; }
        *n_pts = hc;
    }
    *out = m;
    return WASS_OK;
}

int wass_triangulate(wass_ctx* c, const float* disp, int W, int H, const int roi_l[4], const int roi_r[4], const wass_geom* g,
                     const uint8_t* right_img, int img_w, int img_h, const uint8_t* lmask, const uint8_t* rmask,
                     const wass_tri_params* tp, wass_mesh** out, uint64_t* n_pts)
{
    if (!c || !disp || !roi_r || !right_img) return set_err(c, WASS_ERR_INVALID_ARG, "null argument");
    WASS_HIP(c, hipSetDevice(c->device));
    const size_t nd = (size_t)roi_r[2] * roi_r[3], ni = (size_t)img_w * img_h;
    int rc;
    if ((rc = ensure(c, c->fC, nd * 4)) || (rc = ensure(c, c->tmp_in0, ni)) ||
        (lmask && (rc = ensure(c, c->tmp_in1, ni))) || (rmask && (rc = ensure(c, c->tmp_mask, ni))))
        return rc;
    WASS_HIP(c, hipMemcpyAsync(c->fC.p, disp, nd * 4, hipMemcpyHostToDevice, c->ts()));
    WASS_HIP(c, hipMemcpyAsync(c->tmp_in0.p, right_img, ni, hipMemcpyHostToDevice, c->ts()));
    if (lmask) WASS_HIP(c, hipMemcpyAsync(c->tmp_in1.p, lmask, ni, hipMemcpyHostToDevice, c->ts()));
    if (rmask) WASS_HIP(c, hipMemcpyAsync(c->tmp_mask.p, rmask, ni, hipMemcpyHostToDevice, c->ts()));
    return wass_triangulate_dev(c, (const float*)c->fC.p, W, H, roi_l, roi_r, g, (const uint8_t*)c->tmp_in0.p, img_w, img_h,
                                lmask ? (const uint8_t*)c->tmp_in1.p : nullptr, rmask ? (const uint8_t*)c->tmp_mask.p : nullptr,
                                tp, out, n_pts);
}

int wass_mesh_download(wass_ctx* c, const wass_mesh* m, uint8_t* valid, double* p3d, uint8_t* gray)
{
    if (!c || !m) return set_err(c, WASS_ERR_INVALID_ARG, "null argument");
    WASS_HIP(c, hipSetDevice(c->device));
    const size_t n = m->n();
    if (valid) WASS_HIP(c, hipMemcpyAsync(valid, m->valid, n, hipMemcpyDeviceToHost, c->ts()));
    if (gray) WASS_HIP(c, hipMemcpyAsync(gray, m->gray, n, hipMemcpyDeviceToHost, c->ts()));
    if (p3d) {
        int rc = ensure(c, c->scratch, n * 24);
        if (rc) return rc;
        hipLaunchKernelGGL(k_interleave, dim3(nblk(n)), dim3(256), 0, c->ts(), m->x, m->y, m->z, n, (double*)c->scratch.p);
        WASS_HIP(c, hipMemcpyAsync(p3d, c->scratch.p, n * 24, hipMemcpyDeviceToHost, c->ts()));
    }
    WASS_HIP(c, hipStreamSynchronize(c->ts()));
    return WASS_OK;
}

int wass_mesh_upload(wass_ctx* c, int width, int height, const uint8_t* valid, const double* p3d, const uint8_t* gray,
                     wass_mesh** out)
{
    if (!c || !valid || !p3d || !out || width <= 0 || height <= 0) return set_err(c, WASS_ERR_INVALID_ARG, "bad argument");
    WASS_HIP(c, hipSetDevice(c->device));
    wass_mesh* m = nullptr;
    int rc = mesh_alloc(c, width, height, &m);
    if (rc) return rc;
    const size_t n = m->n();
    if ((rc = ensure(c, c->scratch, n * 24))) { wass_mesh_destroy(m); return rc; }
    hipError_t e = hipMemcpyAsync(m->valid, valid, n, hipMemcpyHostToDevice, c->ts());
    if (e == hipSuccess) e = gray ? hipMemcpyAsync(m->gray, gray, n, hipMemcpyHostToDevice, c->ts())
                                  : hipMemsetAsync(m->gray, 0, n, c->ts());
    if (e == hipSuccess) e = hipMemsetAsync(m->codes, 0, n, c->ts());          // an uploaded mesh has no triangulation history
    if (e == hipSuccess) e = hipMemcpyAsync(c->scratch.p, p3d, n * 24, hipMemcpyHostToDevice, c->ts());
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_deinterleave, dim3(nblk(n)), dim3(256), 0, c->ts(), (const double*)c->scratch.p, n, m->x, m->y, m->z);
        e = hipStreamSynchronize(c->ts());
    }
    if (e != hipSuccess) { wass_mesh_destroy(m); return set_err(c, WASS_ERR_DEVICE, "mesh upload: %s", hipGetErrorString(e)); }
    *out = m;
    return WASS_OK;
}

int wass_mesh_zgap_percentile(wass_ctx* c, wass_mesh* m, double percentile, double* out, uint64_t* n_gaps)
{
    if (!c || !m || !out) return set_err(c, WASS_ERR_INVALID_ARG, "null argument");
    WASS_HIP(c, hipSetDevice(c->device));
    const size_t n = m->n(), ng = n * 3;
    int rc = ensure(c, c->scratch, ng * 8 + 2048 * 4 + 64);
    if (rc) return rc;
    unsigned long long* gaps = (unsigned long long*)c->scratch.p;
    unsigned int* hist = (unsigned int*)(gaps + ng);
    unsigned long long* cnt = nullptr;
    if ((rc = counters_reset(c, &cnt))) return rc;
    hipLaunchKernelGGL(k_zgaps, dim3((m->w + 255) / 256, m->h), dim3(256), 0, c->ts(), m->valid, m->z, m->w, m->h, gaps, cnt);
    unsigned long long total = 0;
    if ((rc = counters_sum(c, cnt, &total))) return rc;
    if (n_gaps) *n_gaps = total;
    if (total == 0) { *out = NAN; return WASS_OK; }
    // zgaps[floor(p/100 * n)] of the sorted list (:924), index clamped to n-1
    unsigned long long k = (unsigned long long)floor(percentile / 100.0 * (double)total);
    if (k >= total) k = total - 1;
    unsigned long long prefix = 0;
    std::vector<unsigned int> hh(2048);
    int hi_shift = 64;
    for (int pass = 0; pass < 6; ++pass) {
        const int shift = pass < 5 ? 64 - 11 * (pass + 1) : 0;      // 53,42,31,20,9,0 (last digit: 9 bits)
        WASS_HIP(c, hipMemsetAsync(hist, 0, 2048 * 4, c->ts()));
        hipLaunchKernelGGL(k_radix_hist, dim3(1024), dim3(256), 0, c->ts(), (const unsigned long long*)gaps, ng, shift,
                           pass < 5 ? 2047u : 511u, hi_shift, prefix, hist);
        WASS_HIP(c, hipMemcpyAsync(hh.data(), hist, 2048 * 4, hipMemcpyDeviceToHost, c->ts()));
        WASS_HIP(c, hipStreamSynchronize(c->ts()));
        const int nb = pass < 5 ? 2048 : 512;
        int b = 0;
        for (; b < nb; ++b) { if (k < hh[b]) break; k -= hh[b]; }
        if (b >= nb) return set_err(c, WASS_ERR_DEVICE, "radix select lost its rank (internal error)");
        prefix = (pass < 5 ? (prefix << 11) : (prefix << 9)) | (unsigned long long)b;
        hi_shift = shift;
    }
    double r;
    memcpy(&r, &prefix, 8);
    *out = r;
    return WASS_OK;
}

// the device-resident scalar record of this context
// RT_from_plane (PovMesh.cpp:1044-1069), one body for the host entry point and the device-side frame tail
__host__ __device__ inline void rt_from_plane(const double plane[4], double R[9], double T[3], double Rinv[9], double Tinv[3])
{
    const double a = plane[0], b = plane[1], cc = plane[2], d = plane[3];
    const double q = (1 - cc) / (a * a + b * b);
    R[0] = 1 - a * a * q; R[1] = -a * b * q; R[2] = -a;
    R[3] = -a * b * q; R[4] = 1 - b * b * q; R[5] = -b;
    R[6] = a; R[7] = b; R[8] = cc;
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) Rinv[i * 3 + j] = R[j * 3 + i];
    T[0] = 0; T[1] = 0; T[2] = d;
    const double mT[3] = { -T[0], -T[1], -T[2] };
    for (int i = 0; i < 3; i++) Tinv[i] = Rinv[i * 3] * mT[0] + Rinv[i * 3 + 1] * mT[1] + Rinv[i * 3 + 2] * mT[2];
}

// frame tail, step 1: the plane that main() would pass to save_as_xyz_compressed (wass_stereo.cpp:2108-2123)
__global__ void k_frame_rt(DevState* __restrict__ ds)
{
    if (threadIdx.x || blockIdx.x) return;
    const bool have = ds->ransac_found && ds->refine_ok;
    ds->have_plane = have ? 1u : 0u;
    if (have) {
        rt_from_plane(ds->plane, ds->rtR, ds->rtT, ds->rtRinv, ds->rtTinv);
    } else {
        for (int i = 0; i < 9; ++i) ds->rtR[i] = ds->rtRinv[i] = (i % 4 == 0) ? 1.0 : 0.0;
        for (int i = 0; i < 3; ++i) ds->rtT[i] = ds->rtTinv[i] = 0.0;
    }
}
// frame tail, step 2 -- three passes over the points in one: crop_plane by the refined plane (k_crop_plane_dev), the
// limits of the transformed survivors (k_xyzc_limits_dev) and the per-block survivor counts of the compaction
// (k_block_counts).  Block b owns points [256 b, 256 b + 256), as the pack kernel does.
constexpr int CLC_CHUNKS = 16;
__global__ void __launch_bounds__(256) k_crop_limits_counts_dev(uint8_t* __restrict__ valid, const double* __restrict__ X,
                                                                const double* __restrict__ Y, const double* __restrict__ Z, size_t n,
                                                                const DevState* __restrict__ ds, double thr,
                                                                unsigned long long* __restrict__ kept, unsigned long long* __restrict__ lim,
                                                                unsigned int* __restrict__ blockcnt, unsigned int nchunks)
{
    const bool crop = ds->refine_ok != 0;
    const double a = ds->plane[0], b = ds->plane[1], c = ds->plane[2], d = ds->plane[3];
    double R[9], T[3];
    for (int k = 0; k < 9; ++k) R[k] = ds->rtR[k];
    for (int k = 0; k < 3; ++k) T[k] = ds->rtT[k];
    unsigned long long mn[3] = { ~0ull, ~0ull, ~0ull }, mx[3] = { 0, 0, 0 };
    unsigned int nkept = 0;
    // a workgroup walks CLC_CHUNKS chunks of 256 points (chunk = compaction block of the pack kernel) and meets the global
    // counters once at the end: one update per wave and chunk was half a million same-line atomics
    for (int ch = 0; ch < CLC_CHUNKS; ++ch) {
        const unsigned int chunk = blockIdx.x * CLC_CHUNKS + ch;
        if (chunk >= nchunks) break;
        const size_t i = (size_t)chunk * 256 + threadIdx.x;
        bool v = i < n && valid[i];
        if (v) {
            const double p[3] = { X[i], Y[i], Z[i] };
            if (crop && !(fabs((a * p[0] + b * p[1] + c * p[2]) + d) < thr)) { valid[i] = 0; v = false; }
            if (v) {
                double t[3];
                mulv(R, p, t);
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const unsigned long long key = dkey(t[k] + T[k]);
                    mn[k] = key < mn[k] ? key : mn[k];
                    mx[k] = key > mx[k] ? key : mx[k];
                }
                ++nkept;
            }
        }
        const int cnt = __syncthreads_count(v);
        if (threadIdx.x == 0) blockcnt[chunk] = (unsigned)cnt;
    }
    for (int o = 32; o > 0; o >>= 1) nkept += __shfl_down(nkept, o);
#pragma unroll
    for (int k = 0; k < 3; ++k)
        for (int o = 32; o > 0; o >>= 1) {
            const unsigned long long lo = __shfl_down(mn[k], o), hi = __shfl_down(mx[k], o);
            mn[k] = lo < mn[k] ? lo : mn[k];
            mx[k] = hi > mx[k] ? hi : mx[k];
        }
    if ((threadIdx.x & 63) == 0 && nkept) {
        if (crop) atomicAdd(kept + slot_of_block(), (unsigned long long)nkept);
        unsigned long long* l = lim + (size_t)slot_of_block() * 6;
#pragma unroll
        for (int k = 0; k < 3; ++k) { atomicMin(&l[k], mn[k]); atomicMax(&l[3 + k], mx[k]); }
    }
}
// frame tail: limits -> scale factors, and the 148-byte header of the file image (PovMesh.cpp:417-436)
__global__ void k_frame_header(DevState* __restrict__ ds, const unsigned long long* __restrict__ lim, const unsigned int* __restrict__ total,
                               const unsigned long long* __restrict__ kept /* [2][NSLOT] */, unsigned char* __restrict__ img,
                               const unsigned long long* __restrict__ tri /* [NSLOT] or null */, const unsigned int* __restrict__ inl_total /* or null */,
                               const unsigned int* __restrict__ text_total /* [0] bytes, [1] numbers not formatted; or null */)
{
    if (blockIdx.x) return;
    if (threadIdx.x == 0) {
        const bool have = text_total && inl_total && ds->ransac_found;
        ds->inl_text_bytes = have ? text_total[0] : 0ull;
        ds->inl_text_bad = have ? text_total[1] : 0u;
    }
    {
        unsigned long long k1 = 0, k2 = 0, k3 = 0;
        for (int i = threadIdx.x; i < NSLOT; i += 64) { k1 += kept[i]; k2 += kept[NSLOT + i]; k3 += tri ? tri[i] : 0ull; }
        for (int o = 32; o > 0; o >>= 1) { k1 += __shfl_down(k1, o); k2 += __shfl_down(k2, o); k3 += __shfl_down(k3, o); }
        if (threadIdx.x == 0) { ds->kept1 = k1; ds->kept2 = k2; ds->ntri = k3; ds->ninl_sel = inl_total && ds->ransac_found ? *inl_total : 0u; }
    }
    unsigned long long hl[6] = { ~0ull, ~0ull, ~0ull, 0, 0, 0 };
    for (int i = threadIdx.x; i < NSLOT; i += 64)
        for (int k = 0; k < 3; ++k) {
            if (lim[i * 6 + k] < hl[k]) hl[k] = lim[i * 6 + k];
            if (lim[i * 6 + 3 + k] > hl[3 + k]) hl[3 + k] = lim[i * 6 + 3 + k];
        }
    for (int o = 32; o > 0; o >>= 1)
        for (int k = 0; k < 3; ++k) {
            const unsigned long long a = __shfl_down(hl[k], o), b = __shfl_down(hl[3 + k], o);
            if (a < hl[k]) hl[k] = a;
            if (b > hl[3 + k]) hl[3 + k] = b;
        }
    if (threadIdx.x) return;
    const unsigned int npts = *total;
    double mx[3];
    for (int k = 0; k < 3; ++k) {
        ds->mn[k] = npts ? dunkey(hl[k]) : 1.7976931348623157e308;
        mx[k] = npts ? dunkey(hl[3 + k]) : -1.7976931348623157e308;
        ds->sc[k] = 65535.0 / (mx[k] - ds->mn[k]);
    }
    ds->npts = npts;
    // header: u32 n, f64 scale[3], f64 min[3], f64 Rinv[9], f64 Tinv[3] -- doubles sit at offset 4 (mod 8): word stores
    unsigned int* w = (unsigned int*)img;
    w[0] = npts;
    auto put = [&](int word, double v) {
        const unsigned long long b = (unsigned long long)__double_as_longlong(v);
        w[word] = (unsigned int)b; w[word + 1] = (unsigned int)(b >> 32);
    };
    for (int k = 0; k < 3; ++k) put(1 + 2 * k, ds->sc[k]);
    for (int k = 0; k < 3; ++k) put(7 + 2 * k, ds->mn[k]);
    for (int k = 0; k < 9; ++k) put(13 + 2 * k, ds->rtRinv[k]);
    for (int k = 0; k < 3; ++k) put(31 + 2 * k, ds->rtTinv[k]);
}
__global__ void __launch_bounds__(256) k_xyzc_pack_dev(const uint8_t* __restrict__ valid, const double* __restrict__ X,
                                                       const double* __restrict__ Y, const double* __restrict__ Z, size_t n,
                                                       const DevState* __restrict__ ds, const unsigned int* __restrict__ blockoff,
                                                       uint16_t* __restrict__ out)
{
    RTDev rt;
    for (int i = 0; i < 9; ++i) rt.R[i] = ds->rtR[i];
    for (int i = 0; i < 3; ++i) rt.T[i] = ds->rtT[i];
    xyzc_pack_body(valid, X, Y, Z, n, rt, ds->mn[0], ds->mn[1], ds->mn[2], ds->sc[0], ds->sc[1], ds->sc[2], blockoff, out);
}

constexpr size_t DSTATE_HIST_OFF = (sizeof(DevState) + 255) & ~(size_t)255;
constexpr size_t DSTATE_STRIDE = (DSTATE_HIST_OFF + (size_t)GAP_HIST_COPIES * GAP_BINS * 4 + 255) & ~(size_t)255;
// two records (+ histograms), alternated by the frame tail (wass_ctx::ds_slot): frame n's is downloaded while frame n+1's is being built
static int dstate(wass_ctx* c, DevState** ds)
{
    int rc = ensure(c, c->dstate, 2 * DSTATE_STRIDE);
    if (rc) return rc;
    *ds = (DevState*)((char*)c->dstate.p + (size_t)(c->ds_slot & 1) * DSTATE_STRIDE);
    return WASS_OK;
}

// connected components with the z gap taken from device memory; leaves (size << 32 | ~mincm) in ds->ccl_best
static int enqueue_ccl(wass_ctx* c, wass_mesh* m, DevState* ds)
{
    const size_t n = m->n();
    if (n > 0x7FFFFFFFull) return set_err(c, WASS_ERR_UNSUPPORTED, "mesh too large");
    int rc = ensure(c, c->scratch, n * 12);
    if (rc) return rc;
    int* parent = (int*)c->scratch.p;
    unsigned int* size = (unsigned int*)(parent + n);
    unsigned int* mincm = size + n;
    unsigned long long* best = &ds->ccl_best;
    const dim3 blk(256), g1(nblk(n));
    hipLaunchKernelGGL(k_ccl_init, g1, blk, 0, c->ts(), m->valid, m->z, m->w, (int)n, (const double*)&ds->zgap, parent, size, mincm, best);
    hipLaunchKernelGGL(k_ccl_merge, g1, blk, 0, c->ts(), m->valid, m->z, m->w, (int)n, (const double*)&ds->zgap, parent);
    hipLaunchKernelGGL(k_ccl_flatten, g1, blk, 0, c->ts(), (int)n, parent);
    hipLaunchKernelGGL(k_ccl_count, dim3((unsigned)((n + 1024 * CCL_COUNT_CHUNKS - 1) / (1024 * CCL_COUNT_CHUNKS))), dim3(1024), 0, c->ts(), (int)n, m->w, m->h,
                       (const int*)parent, size, mincm);
    hipLaunchKernelGGL(k_ccl_best, g1, blk, 0, c->ts(), (int)n, (const int*)parent, (const unsigned int*)size,
                       (const unsigned int*)mincm, best);
    // no valid point at all: best stays 0 and no root key can equal it (sizes are >= 1) -> nothing kept, as in the
    // reference where extract_component(0) then matches no point
    hipLaunchKernelGGL(k_ccl_keep, g1, blk, 0, c->ts(), m->valid, (int)n, (const int*)parent, (const unsigned int*)size,
                       (const unsigned int*)mincm, (const unsigned long long*)best);
    WASS_HIP(c, hipGetLastError());
    return WASS_OK;
}

int wass_mesh_keep_biggest_component(wass_ctx* c, wass_mesh* m, double zgap, uint64_t* size_out)
{
    if (!c || !m) return set_err(c, WASS_ERR_INVALID_ARG, "null argument");
    WASS_HIP(c, hipSetDevice(c->device));
    DevState* ds = nullptr;
    int rc = dstate(c, &ds);
    if (rc) return rc;
    WASS_HIP(c, hipMemcpyAsync(&ds->zgap, &zgap, 8, hipMemcpyHostToDevice, c->ts()));
    WASS_HIP(c, hipStreamSynchronize(c->ts()));       // zgap lives on the caller's stack
    if ((rc = enqueue_ccl(c, m, ds))) return rc;
    if (size_out) {
        unsigned long long hb = 0;
        WASS_HIP(c, hipMemcpyAsync(&hb, &ds->ccl_best, 8, hipMemcpyDeviceToHost, c->ts()));
        WASS_HIP(c, hipStreamSynchronize(c->ts()));
        *size_out = hb >> 32;
    }
    return WASS_OK;
}

// wass_stereo.cpp:2046-2050 as one call: compute_zgap_percentile + cluster_biggest_connected_component with every
// intermediate decision taken on the device (6 radix-select passes, component choice) and one read-back at the end.
// Pinned, context-owned source images of the small host-to-device copies of the sync-free frame tail.  An asynchronous
// copy from pageable or stack memory is only safe if the runtime happens to stage it before returning; these copies
// are enqueued and never waited for, so their sources must outlive the call: [DevState init | limits init | uv samples].
constexpr size_t STAGE_UV_OFF = 4096 + NSLOT * 6 * 8;
constexpr size_t STAGE_UV_BYTES = 1800 * 24;                  // PLANE_RANSAC_ROUNDS <= 1800 (LDS limit of k_ransac_score)
constexpr size_t STAGE_BYTES = STAGE_UV_OFF + 2 * STAGE_UV_BYTES;   // two areas, alternated: the host never waits for the previous frame's copy
static int host_stage(wass_ctx* c, unsigned char** out)
{
    static_assert(sizeof(DevState) <= 4096, "DevState init image");
    if (!c->h_stage) {
        if (hipHostMalloc((void**)&c->h_stage, STAGE_BYTES, hipHostMallocDefault) != hipSuccess)
            return set_err(c, WASS_ERR_NO_MEMORY, "hipHostMalloc failed");
        unsigned char* p = (unsigned char*)c->h_stage;
        DevState init;
        memset(&init, 0, sizeof init);
        init.sel_hi_shift = 64;
        memcpy(p, &init, sizeof init);
        unsigned long long* lim = (unsigned long long*)(p + 4096);
        for (int i = 0; i < NSLOT; ++i) for (int k = 0; k < 6; ++k) lim[i * 6 + k] = k < 3 ? ~0ull : 0ull;
        if (hipEventCreateWithFlags(&c->ev_stage, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&c->ev_stage2, hipEventDisableTiming) != hipSuccess)
            return set_err(c, WASS_ERR_DEVICE, "hipEventCreate failed");
    }
    *out = (unsigned char*)c->h_stage;
    return WASS_OK;
}

static int enqueue_remove_outliers(wass_ctx* c, wass_mesh* m, double percentile, DevState** dsp)
{
    DevState* ds = nullptr;
    int rc = dstate(c, &ds);
    if (rc) return rc;
    unsigned int* hist = (unsigned int*)((char*)ds + DSTATE_HIST_OFF);      // 256-byte aligned: one fill kernel, not three
    unsigned char* stage = nullptr;
    if ((rc = host_stage(c, &stage))) return rc;
    WASS_HIP(c, hipStreamWaitEvent(c->ts(), c->fslot[c->ds_slot & 1].ev_copy, 0));   // the download that read THIS record: two frames ago
    WASS_HIP(c, hipMemcpyAsync(ds, stage, sizeof(DevState), hipMemcpyHostToDevice, c->ts()));
    WASS_HIP(c, hipMemsetAsync(hist, 0, (size_t)GAP_HIST_COPIES * GAP_BINS * 4, c->ts()));
    for (int pass = 0; pass < GAP_PASSES; ++pass) {
        const int shift = pass < GAP_PASSES - 1 ? 64 - GAP_BITS * (pass + 1) : 0;       // 53, 42, 31, 20, 9, 0 (last digit: 9 bits)
        const int nbits = pass < GAP_PASSES - 1 ? GAP_BITS : 64 - GAP_BITS * (GAP_PASSES - 1);
        hipLaunchKernelGGL(k_gap_hist, dim3((m->w + 255) / 256, 256), dim3(256), 0, c->ts(), m->valid, m->z, m->w, m->h, shift, (1u << nbits) - 1u,
                           (const DevState*)ds, hist);
        hipLaunchKernelGGL(k_radix_pick, dim3(1), dim3(256), 0, c->ts(), hist, pass, shift, nbits, percentile, ds);
    }
    if (c->ev_tail[2]) (void)hipEventRecord(c->ev_tail[2], c->ts());
    if ((rc = enqueue_ccl(c, m, ds))) return rc;
    if (c->ev_tail[3]) (void)hipEventRecord(c->ev_tail[3], c->ts());
    *dsp = ds;
    return WASS_OK;
}

int wass_mesh_remove_outliers(wass_ctx* c, wass_mesh* m, double percentile, double* zgap_out, uint64_t* n_gaps, uint64_t* size_out)
{
    if (!c || !m) return set_err(c, WASS_ERR_INVALID_ARG, "null argument");
    WASS_HIP(c, hipSetDevice(c->device));
    DevState* ds = nullptr;
    int rc = enqueue_remove_outliers(c, m, percentile, &ds);
    if (rc) return rc;
    DevState h;
    WASS_HIP(c, hipMemcpyAsync(&h, ds, sizeof h, hipMemcpyDeviceToHost, c->ts()));
    WASS_HIP(c, hipStreamSynchronize(c->ts()));
    if (h.sel_fail == 2) return set_err(c, WASS_ERR_DEVICE, "radix select lost its rank (internal error)");
    if (zgap_out) *zgap_out = h.zgap;
    if (n_gaps) *n_gaps = h.sel_total;
    if (size_out) *size_out = h.ccl_best >> 32;
    return WASS_OK;
}

}  // extern "C"

// PovMesh.cpp:680-691: three (u, v) grid positions per round, re-drawn while any two are closer than 1 % of the height.
// GCC evaluates the arguments of cv::Vec2i(rand()%iW, rand()%iH) right to left: v is drawn first.
template <typename Rand>
static int ransac_sample_with(Rand&& next, int width, int height, int rounds, int32_t* uv)
{
    if (width <= 0 || height <= 0 || rounds < 0 || !uv) return WASS_ERR_INVALID_ARG;
    const double mindist = height * 0.01;
    int r = 0;
    long guard = 0;
    while (r < rounds) {
        int p[6];
        for (int k = 0; k < 3; k++) { const int v = next() % height; const int u = next() % width; p[2 * k] = u; p[2 * k + 1] = v; }
        const double d12 = sqrt((double)(p[0] - p[2]) * (p[0] - p[2]) + (double)(p[1] - p[3]) * (p[1] - p[3]));
        const double d23 = sqrt((double)(p[2] - p[4]) * (p[2] - p[4]) + (double)(p[3] - p[5]) * (p[3] - p[5]));
        const double d13 = sqrt((double)(p[0] - p[4]) * (p[0] - p[4]) + (double)(p[1] - p[5]) * (p[1] - p[5]));
        if (d12 < mindist || d23 < mindist || d13 < mindist) {
            if (++guard > 100000000L) return WASS_ERR_INVALID_ARG;   // degenerate grid (the reference would spin forever)
            continue;
        }
        for (int k = 0; k < 6; k++) uv[(size_t)r * 6 + k] = p[k];
        r++;
    }
    return WASS_OK;
}

// glibc's srand(seed) + rand() restated (the TYPE_3 additive feedback generator of stdlib/random_r.c: 31 words seeded
// by the Lehmer recurrence 16807 x mod 2^31 - 1, taps 31 and 3, the first 310 outputs discarded, result = word >> 1),
// with PRIVATE state: the sequence the reference sees when RANSAC is the only consumer of rand() after srand(seed),
// which is the case in wass_stereo (PovMesh.cpp:680-682 holds its only rand() calls).
namespace {
struct GlibcRand {
    uint32_t r[34];
    int i = 0;
    explicit GlibcRand(uint32_t seed)
    {
        int32_t w[34];
        w[0] = seed == 0 ? 1 : (int32_t)seed;
        for (int k = 1; k < 31; ++k) {
            // 16807 * w mod (2^31 - 1) by Schrage's split, exactly as glibc does it (the first word may be negative)
            const long hi = w[k - 1] / 127773, lo = w[k - 1] % 127773;
            long word = 16807 * lo - 2836 * hi;
            if (word < 0) word += 2147483647;
            w[k] = (int32_t)word;
        }
        for (int k = 0; k < 31; ++k) r[k] = (uint32_t)w[k];
        for (int k = 31; k < 34; ++k) r[k] = r[k - 31];
        // ring of 34 words: word n = word n-31 + word n-3; glibc discards 310 outputs after seeding
        i = 0;
        for (int k = 0; k < 310; ++k) (void)next_word();
    }
    uint32_t next_word()
    {
        // position i holds word n-34; n-31 is i+3, n-3 is i+31
        const uint32_t v = r[(i + 3) % 34] + r[(i + 31) % 34];
        r[i] = v;
        i = (i + 1) % 34;
        return v;
    }
    int operator()() { return (int)(next_word() >> 1); }
};
}  // namespace

extern "C" {

// libc rand(); the caller has srand()'ed (wass_stereo.cpp:1864-1872).  rand() has process-wide state: any other consumer
// between srand() and this call (measured: threads of the HIP runtime draw from it) changes the sequence -- a host that
// wants the reference's run-to-run reproducibility for a fixed RANDOM_SEED uses wass_ransac_sample_seeded.
int wass_ransac_sample(int width, int height, int rounds, int32_t* uv)
{
    return ransac_sample_with([]() { return rand(); }, width, height, rounds, uv);
}

int wass_ransac_sample_seeded(uint32_t seed, int width, int height, int rounds, int32_t* uv)
{
    GlibcRand g(seed);
    return ransac_sample_with(g, width, height, rounds, uv);
}

int wass_mesh_ransac_plane(wass_ctx* c, wass_mesh* m, const int32_t* uv, int rounds, double thr, double plane_out[4],
                           uint64_t* best_inliers, int* found)
{
    if (!c || !m || !uv || !plane_out || rounds <= 0) return set_err(c, WASS_ERR_INVALID_ARG, "bad argument");
    for (int r = 0; r < rounds * 3; ++r)
        if (uv[2 * r] < 0 || uv[2 * r] >= m->w || uv[2 * r + 1] < 0 || uv[2 * r + 1] >= m->h)
            return set_err(c, WASS_ERR_INVALID_ARG, "sample %d outside the mesh grid", r / 3);
    WASS_HIP(c, hipSetDevice(c->device));
    const size_t n = m->n();
    const size_t bytes = (size_t)rounds * (24 + sizeof(PlaneCand) + 8) + 256;
    int rc = ensure(c, c->scratch, bytes);
    if (rc) return rc;
    PlaneCand* cand = (PlaneCand*)c->scratch.p;
    unsigned long long* counts = (unsigned long long*)(cand + rounds);
    int32_t* duv = (int32_t*)(counts + rounds);
    WASS_HIP(c, hipMemcpyAsync(duv, uv, (size_t)rounds * 24, hipMemcpyHostToDevice, c->ts()));
    hipLaunchKernelGGL(k_ransac_planes, dim3((rounds + 63) / 64), dim3(64), 0, c->ts(), m->valid, m->x, m->y, m->z, m->w,
                       (const int32_t*)duv, rounds, cand, counts, (unsigned long long*)nullptr, 0);
    constexpr int PTS = 8;
    const size_t lds = (size_t)rounds * (32 + 4);
    if (lds > 64 * 1024) return set_err(c, WASS_ERR_UNSUPPORTED, "PLANE_RANSAC_ROUNDS %d too large (max 1800)", rounds);
    hipLaunchKernelGGL(k_ransac_score<PTS>, dim3((m->w + 255) / 256, (m->h + PTS - 1) / PTS), dim3(256), lds, c->ts(),
                       m->valid, m->x, m->y, m->z, m->w, m->h, (const PlaneCand*)cand, rounds, thr, counts);
    std::vector<PlaneCand> hc(rounds);
    std::vector<unsigned long long> hn(rounds);
    WASS_HIP(c, hipMemcpyAsync(hc.data(), cand, (size_t)rounds * sizeof(PlaneCand), hipMemcpyDeviceToHost, c->ts()));
    WASS_HIP(c, hipMemcpyAsync(hn.data(), counts, (size_t)rounds * 8, hipMemcpyDeviceToHost, c->ts()));
    WASS_HIP(c, hipStreamSynchronize(c->ts()));
    unsigned long long best = 0;
    double bn[3] = { 0, 0, 0 }, bd = 0;
    for (int r = 0; r < rounds; ++r)          // first strictly better candidate wins (:750-755)
        if (hc[r].ok && hn[r] > best) { best = hn[r]; bn[0] = hc[r].n[0]; bn[1] = hc[r].n[1]; bn[2] = hc[r].n[2]; bd = hc[r].d; }
    plane_out[0] = bn[0]; plane_out[1] = bn[1]; plane_out[2] = bn[2]; plane_out[3] = bd;
    if (best_inliers) *best_inliers = best;
    if (found) *found = best < n / 10 ? 0 : 1;  // :773
    return WASS_OK;
}

int wass_mesh_crop_plane(wass_ctx* c, wass_mesh* m, const double plane[4], double thr, uint64_t* kept)
{
    if (!c || !m || !plane) return set_err(c, WASS_ERR_INVALID_ARG, "null argument");
    WASS_HIP(c, hipSetDevice(c->device));
    unsigned long long* cnt = nullptr;
    int rc = counters_reset(c, &cnt);
    if (rc) return rc;
    hipLaunchKernelGGL(k_crop_plane, dim3(2048), dim3(256), 0, c->ts(), m->valid, m->x, m->y, m->z, m->n(), plane[0],
                       plane[1], plane[2], plane[3], thr, cnt);
    unsigned long long hk = 0;
    if ((rc = counters_sum(c, cnt, &hk))) return rc;
    if (kept) *kept = hk;
    return WASS_OK;
}

int wass_mesh_refine_plane(wass_ctx* c, wass_mesh* m, const wass_refine_params* rp, double plane_out[4], uint64_t* n_inliers)
{
    if (!c || !m || !rp || !plane_out) return set_err(c, WASS_ERR_INVALID_ARG, "null argument");
    WASS_HIP(c, hipSetDevice(c->device));
    RefineDev rd;
    rd.xmin = rp->xmin; rd.xmax = rp->xmax; rd.ymin = rp->ymin; rd.ymax = rp->ymax; rd.maxd = rp->max_distance;
    rd.weighted = rp->weight_by_distance;
    rd.umin = rp->central_third_only ? m->w / 4 : 0;            // :585-588
    rd.umax = rp->central_third_only ? m->w * 3 / 4 : m->w - 1;
    rd.vmin = rp->central_third_only ? m->h / 4 : 0;
    rd.vmax = rp->central_third_only ? m->h * 2 / 3 : m->h - 1;
    const int NB = 1024;
    int rc = ensure(c, c->scratch, (size_t)NB * 6 * 8);
    if (rc) return rc;
    double* part = (double*)c->scratch.p;
    std::vector<double> hp((size_t)NB * 6);
    hipLaunchKernelGGL(k_refine_moments, dim3(NB), dim3(256), 0, c->ts(), m->valid, m->x, m->y, m->z, m->w, m->n(), rd, part);
    WASS_HIP(c, hipMemcpyAsync(hp.data(), part, (size_t)NB * 5 * 8, hipMemcpyDeviceToHost, c->ts()));
    WASS_HIP(c, hipStreamSynchronize(c->ts()));
    double mom[5];
    sum_partials_host<5>(hp.data(), NB, mom);
    if (n_inliers) *n_inliers = (uint64_t)(mom[0] + 0.5);
    if (mom[0] < 3 || !(mom[1] > 0)) return set_err(c, WASS_ERR_TOO_FEW_POINTS, "plane refinement has %g inliers", mom[0]);
    const double cx = mom[2] / mom[1], cy = mom[3] / mom[1], cz = mom[4] / mom[1];
    hipLaunchKernelGGL(k_refine_cov, dim3(NB), dim3(256), 0, c->ts(), m->valid, m->x, m->y, m->z, m->w, m->n(), rd, cx, cy, cz,
                       part);
    WASS_HIP(c, hipMemcpyAsync(hp.data(), part, (size_t)NB * 6 * 8, hipMemcpyDeviceToHost, c->ts()));
    WASS_HIP(c, hipStreamSynchronize(c->ts()));
    double s[6];
    sum_partials_host<6>(hp.data(), NB, s);
    const double A[9] = { s[0], s[1], s[2], s[1], s[3], s[4], s[2], s[4], s[5] };
    double nrm[3];
    smallest_eigvec3(A, nrm);
    const double nn = sqrt(nrm[0] * nrm[0] + nrm[1] * nrm[1] + nrm[2] * nrm[2]);
    nrm[0] /= nn; nrm[1] /= nn; nrm[2] /= nn;
    if (nrm[2] < 0) { nrm[0] *= -1.0; nrm[1] *= -1.0; nrm[2] *= -1.0; }   // :646-649
    plane_out[0] = nrm[0]; plane_out[1] = nrm[1]; plane_out[2] = nrm[2];
    plane_out[3] = -(nrm[0] * cx + nrm[1] * cy + nrm[2] * cz);
    return WASS_OK;
}

int wass_mesh_refinement_inliers(wass_ctx* c, wass_mesh* m, const wass_refine_params* rp, int every, double** xyz_out, uint64_t* n_out)
{
    if (!c || !m || !rp || !xyz_out || !n_out || every <= 0) return set_err(c, WASS_ERR_INVALID_ARG, "bad argument");
    WASS_HIP(c, hipSetDevice(c->device));
    *xyz_out = nullptr; *n_out = 0;
    RefineDev rd;
    rd.xmin = rp->xmin; rd.xmax = rp->xmax; rd.ymin = rp->ymin; rd.ymax = rp->ymax; rd.maxd = rp->max_distance;
    rd.weighted = rp->weight_by_distance;
    rd.umin = rp->central_third_only ? m->w / 4 : 0;
    rd.umax = rp->central_third_only ? m->w * 3 / 4 : m->w - 1;
    rd.vmin = rp->central_third_only ? m->h / 4 : 0;
    rd.vmax = rp->central_third_only ? m->h * 2 / 3 : m->h - 1;
    const size_t n = m->n();
    const unsigned nb = nblk(n);
    const size_t cap = (n + (size_t)every - 1) / (size_t)every;                 // at most every point is an inlier
    const size_t off_cnt = 64, off_out = (off_cnt + (size_t)nb * 4 + 255) & ~(size_t)255;
    int rc = ensure(c, c->scratch, off_out + cap * 24);
    if (rc) return rc;
    unsigned int* total = (unsigned int*)c->scratch.p;
    unsigned int* bcnt = (unsigned int*)((char*)c->scratch.p + off_cnt);
    double* dout = (double*)((char*)c->scratch.p + off_out);
    hipStream_t s = c->ts();
    hipLaunchKernelGGL(k_inlier_counts, dim3(nb), dim3(256), 0, s, m->valid, m->x, m->y, m->z, m->w, n, rd, bcnt);
    hipLaunchKernelGGL(k_scan_blocks, dim3(1), dim3(256), 0, s, bcnt, (int)nb, total);
    hipLaunchKernelGGL(k_inlier_pack, dim3(nb), dim3(256), 0, s, m->valid, m->x, m->y, m->z, m->w, n, rd, (const unsigned int*)bcnt,
                       (unsigned)every, dout);
    unsigned int ht = 0;
    WASS_HIP(c, hipMemcpyAsync(&ht, total, 4, hipMemcpyDeviceToHost, s));
    WASS_HIP(c, hipStreamSynchronize(s));
    const size_t keep = ((size_t)ht + (size_t)every - 1) / (size_t)every;
    if (keep == 0) return WASS_OK;
    double* host = (double*)malloc(keep * 24);
    if (!host) return set_err(c, WASS_ERR_NO_MEMORY, "out of host memory");
    hipError_t e = hipMemcpyAsync(host, dout, keep * 24, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) { free(host); return set_err(c, WASS_ERR_DEVICE, "download: %s", hipGetErrorString(e)); }
    *xyz_out = host; *n_out = keep;
    return WASS_OK;
}

// wass_stereo.cpp:2062-2107 as one call: ransac_find_plane -> crop_plane(ransac_thr) -> refine_plane ->
// crop_plane(max_distance); candidate choice, centroid and the 3x3 eigen-solve run on the device, one read-back.
static int enqueue_fit_plane(wass_ctx* c, wass_mesh* m, const int32_t* uv, int rounds, double ransac_thr, const wass_refine_params* rp,
                             double max_distance, DevState** dsp, unsigned long long** keptp, bool final_crop = true)
{
    if (!c || !m || !uv || !rp || rounds <= 0) return set_err(c, WASS_ERR_INVALID_ARG, "bad argument");
    for (int r = 0; r < rounds * 3; ++r)
        if (uv[2 * r] < 0 || uv[2 * r] >= m->w || uv[2 * r + 1] < 0 || uv[2 * r + 1] >= m->h)
            return set_err(c, WASS_ERR_INVALID_ARG, "sample %d outside the mesh grid", r / 3);
    WASS_HIP(c, hipSetDevice(c->device));
    const size_t n = m->n();
    DevState* ds = nullptr;
    int rc = dstate(c, &ds);
    if (rc) return rc;
    const int NB = 1024;
    const size_t cand_bytes = (((size_t)rounds * (24 + sizeof(PlaneCand) + 8) + 256) + 255) & ~(size_t)255;
    if ((rc = ensure(c, c->scratch, cand_bytes + (size_t)NB * 6 * 8))) return rc;
    if ((rc = ensure(c, c->counters, (size_t)NSLOT * 6 * 8))) return rc;
    PlaneCand* cand = (PlaneCand*)c->scratch.p;
    unsigned long long* counts = (unsigned long long*)(cand + rounds);
    int32_t* duv = (int32_t*)(counts + rounds);
    double* part = (double*)((char*)c->scratch.p + cand_bytes);
    unsigned long long* kept1 = (unsigned long long*)c->counters.p;            // [NSLOT]
    unsigned long long* kept2 = kept1 + NSLOT;
    constexpr int PTS = 8;
    const size_t lds = (size_t)rounds * (32 + 4);
    if (lds > 64 * 1024) return set_err(c, WASS_ERR_UNSUPPORTED, "PLANE_RANSAC_ROUNDS %d too large (max 1800)", rounds);
    hipStream_t s = c->ts();
    {   // the caller's sample array may be pageable and short-lived: go through the pinned stage (the previous frame's
        // copy out of it was enqueued a whole frame ago; wait for it before overwriting)
        unsigned char* stage = nullptr;
        if ((rc = host_stage(c, &stage))) return rc;
        // (two areas, alternated with the frame slot: the copy out of THIS one was enqueued two frames ago)
        const int ua = c->ds_slot & 1;
        hipEvent_t evs = ua ? c->ev_stage2 : c->ev_stage;
        unsigned char* area = stage + STAGE_UV_OFF + (size_t)ua * STAGE_UV_BYTES;
        if (c->stage_uv_busy) WASS_HIP(c, hipEventSynchronize(evs));
        memcpy(area, uv, (size_t)rounds * 24);
        WASS_HIP(c, hipMemcpyAsync(duv, area, (size_t)rounds * 24, hipMemcpyHostToDevice, s));
        WASS_HIP(c, hipEventRecord(evs, s));
        c->stage_uv_busy = true;
    }
    hipLaunchKernelGGL(k_ransac_planes, dim3((rounds + 63) / 64), dim3(64), 0, s, m->valid, m->x, m->y, m->z, m->w, (const int32_t*)duv,
                       rounds, cand, counts, kept1, 2 * NSLOT);
    hipLaunchKernelGGL(k_ransac_score<PTS>, dim3((m->w + 255) / 256, (m->h + PTS - 1) / PTS), dim3(256), lds, s, m->valid, m->x,
                       m->y, m->z, m->w, m->h, (const PlaneCand*)cand, rounds, ransac_thr, counts);
    hipLaunchKernelGGL(k_ransac_pick, dim3(1), dim3(64), 0, s, (const PlaneCand*)cand, (const unsigned long long*)counts, rounds, n, ds);
    if (c->ev_tail[4]) (void)hipEventRecord(c->ev_tail[4], s);
    RefineDev rd;
    rd.xmin = rp->xmin; rd.xmax = rp->xmax; rd.ymin = rp->ymin; rd.ymax = rp->ymax; rd.maxd = rp->max_distance;
    rd.weighted = rp->weight_by_distance;
    rd.umin = rp->central_third_only ? m->w / 4 : 0;
    rd.umax = rp->central_third_only ? m->w * 3 / 4 : m->w - 1;
    rd.vmin = rp->central_third_only ? m->h / 4 : 0;
    rd.vmax = rp->central_third_only ? m->h * 2 / 3 : m->h - 1;
    hipLaunchKernelGGL(k_crop_moments_dev, dim3(NB), dim3(256), 0, s, m->valid, m->x, m->y, m->z, m->w, n, (const double*)ds->ransac_plane,
                       (const int*)&ds->ransac_found, ransac_thr, kept1, rd, part);
    hipLaunchKernelGGL(k_refine_centroid, dim3(1), dim3(64), 0, s, (const double*)part, NB, (const int*)&ds->ransac_found, ds);
    hipLaunchKernelGGL(k_refine_cov_dev, dim3(NB), dim3(256), 0, s, m->valid, m->x, m->y, m->z, m->w, n, rd, (const DevState*)ds, part);
    hipLaunchKernelGGL(k_refine_finish, dim3(1), dim3(64), 0, s, (const double*)part, NB, ds);
    if (final_crop)                                              // the frame tail folds this pass into its limits / counts pass
        hipLaunchKernelGGL(k_crop_plane_dev, dim3(2048), dim3(256), 0, s, m->valid, m->x, m->y, m->z, n, (const double*)ds->plane,
                           (const int*)&ds->refine_ok, max_distance, kept2);
    WASS_HIP(c, hipGetLastError());
    *dsp = ds;
    *keptp = kept1;
    return WASS_OK;
}

int wass_mesh_fit_plane(wass_ctx* c, wass_mesh* m, const int32_t* uv, int rounds, double ransac_thr, const wass_refine_params* rp,
                        double max_distance, wass_plane_result* out)
{
    if (!out) return set_err(c, WASS_ERR_INVALID_ARG, "bad argument");
    DevState* ds = nullptr;
    unsigned long long* kept1 = nullptr;
    int rc = enqueue_fit_plane(c, m, uv, rounds, ransac_thr, rp, max_distance, &ds, &kept1);
    if (rc) return rc;
    hipStream_t s = c->ts();
    DevState h;
    unsigned long long hk[2 * NSLOT];
    WASS_HIP(c, hipMemcpyAsync(&h, ds, sizeof h, hipMemcpyDeviceToHost, s));
    WASS_HIP(c, hipMemcpyAsync(hk, kept1, sizeof hk, hipMemcpyDeviceToHost, s));
    WASS_HIP(c, hipStreamSynchronize(s));
    memset(out, 0, sizeof *out);
    out->found = h.ransac_found;
    out->ransac_inliers = h.ransac_best;
    for (int k = 0; k < 4; ++k) { out->ransac_plane[k] = h.ransac_plane[k]; out->plane[k] = h.ransac_found && h.refine_ok ? h.plane[k] : NAN; }
    if (h.ransac_found) {
        for (int i = 0; i < NSLOT; ++i) { out->kept_after_ransac_crop += hk[i]; out->kept_final += hk[NSLOT + i]; }
        out->refine_inliers = (uint64_t)(h.ninl + 0.5);
        if (!h.refine_ok) return set_err(c, WASS_ERR_TOO_FEW_POINTS, "plane refinement has %g inliers", h.ninl);
    }
    return WASS_OK;
}

// The whole mesh tail of main() (wass_stereo.cpp:2046-2123) enqueued without a single host synchronisation: every
// decision (percentile, component, best candidate, found / not found, refined plane, R|T, limits, point count) is
// taken on the device, the file image (header included) is assembled in HBM and downloaded on the copy stream.
int wass_mesh_finish_frame_async(wass_ctx* c, wass_mesh* m, double percentile, const int32_t* uv, int rounds, double ransac_thr,
                                 const wass_refine_params* rp, double max_distance, void* dst, size_t capacity)
{
    return wass_mesh_finish_frame_async_ex(c, m, percentile, uv, rounds, ransac_thr, rp, max_distance, dst, capacity, nullptr, 0, 0, nullptr);
}

int wass_mesh_finish_frame_async_ex(wass_ctx* c, wass_mesh* m, double percentile, const int32_t* uv, int rounds, double ransac_thr,
                                    const wass_refine_params* rp, double max_distance, void* dst, size_t capacity,
                                    double* inliers_dst, size_t inliers_capacity, int inliers_every, uint8_t* component_mask_dst)
{
    return wass_mesh_finish_frame_async_ex2(c, m, percentile, uv, rounds, ransac_thr, rp, max_distance, dst, capacity, inliers_dst, inliers_capacity,
                                            inliers_every, component_mask_dst, nullptr, 0);
}

int wass_mesh_finish_frame_async_ex2(wass_ctx* c, wass_mesh* m, double percentile, const int32_t* uv, int rounds, double ransac_thr,
                                     const wass_refine_params* rp, double max_distance, void* dst, size_t capacity,
                                     double* inliers_dst, size_t inliers_capacity, int inliers_every, uint8_t* component_mask_dst,
                                     char* inliers_text_dst, size_t inliers_text_capacity)
{
    if (!c || !m || !dst) return set_err(c, WASS_ERR_INVALID_ARG, "null argument");
    if (inliers_dst && (inliers_every <= 0 || inliers_capacity == 0)) return set_err(c, WASS_ERR_INVALID_ARG, "bad inlier selection");
    if (inliers_text_dst && inliers_every <= 0) return set_err(c, WASS_ERR_INVALID_ARG, "bad inlier selection");
    WASS_HIP(c, hipSetDevice(c->device));
    const size_t n = m->n();
    if (capacity < 148 + n * 6)
        return set_err(c, WASS_ERR_INVALID_ARG, "the asynchronous form needs room for every grid point: %zu bytes", 148 + n * 6);
    if (c->nframe_enq - c->nframe_col >= 2)
        return set_err(c, WASS_ERR_INVALID_ARG, "two frames are pending already: read the older one's record first (wass_ctx_frame_result)");
    const int slot = (int)(c->nframe_enq & 1);
    wass_ctx::FrameSlot& fs = c->fslot[slot];
    if (!fs.h_frame && hipHostMalloc((void**)&fs.h_frame, sizeof(DevState) + 64, hipHostMallocDefault) != hipSuccess)
        return set_err(c, WASS_ERR_NO_MEMORY, "hipHostMalloc failed");
    c->ds_slot = slot;                                     // the stage helpers below build THIS slot's record
    wass::Buf& inlbuf = slot ? c->inl2 : c->inl;           // ... and the selection lives in this slot's buffer
    hipEvent_t ev_prev = c->ev_copy;                       // the previous frame's downloads (they read the shared file image and masks)
    DevState* ds = nullptr;
    unsigned long long* kept1 = nullptr;
    int rc;
    const bool timed = c->ev_tail[0] != nullptr;          // a wass_triangulate[_dev] call of this context has created (and recorded) them
    if (timed) (void)hipEventRecord(c->ev_tail[1], c->ts());
    if ((rc = enqueue_remove_outliers(c, m, percentile, &ds))) return rc;
    if (component_mask_dst) {                              // what cluster_biggest_connected_component left valid (graph_components.jpg)
        if ((rc = ensure(c, c->ccmask, n))) return rc;
        WASS_HIP(c, hipStreamWaitEvent(c->ts(), ev_prev, 0));
        WASS_HIP(c, hipMemcpyAsync(c->ccmask.p, m->valid, n, hipMemcpyDeviceToDevice, c->ts()));
    }
    if ((rc = enqueue_fit_plane(c, m, uv, rounds, ransac_thr, rp, max_distance, &ds, &kept1, false))) return rc;
    const unsigned nb = nblk(n);
    if ((rc = ensure(c, c->xyzc, 148 + n * 6 + 16))) return rc;
    // block counts live behind the candidate area of the scratch buffer, which enqueue_fit_plane sized; grow if needed
    const size_t need = 64 + (size_t)nb * 4 + 16;
    if (c->scratch.cap < need && (rc = ensure(c, c->scratch, need))) return rc;
    hipStream_t s = c->ts();
    // plane_refinement_inliers.xyz: the refinement inliers are the points that survived the crop by the RANSAC plane, which
    // enqueue_fit_plane has just applied; the final crop (below) has not run yet -- the point main() collects them at
    size_t inl_copy = 0, text_copy = 0;
    const char* text_src = nullptr;
    const unsigned int* text_totals = nullptr;             // [0] bytes of text, [1] numbers not formatted: written on the copy stream
    unsigned int* inl_total = nullptr;
    if (inliers_dst || inliers_text_dst) {
        RefineDev rd;
        rd.xmin = rp->xmin; rd.xmax = rp->xmax; rd.ymin = rp->ymin; rd.ymax = rp->ymax; rd.maxd = rp->max_distance;
        rd.weighted = rp->weight_by_distance;
        rd.umin = rp->central_third_only ? m->w / 4 : 0;
        rd.umax = rp->central_third_only ? m->w * 3 / 4 : m->w - 1;
        rd.vmin = rp->central_third_only ? m->h / 4 : 0;
        rd.vmax = rp->central_third_only ? m->h * 2 / 3 : m->h - 1;
        const size_t cap = (n + (size_t)inliers_every - 1) / (size_t)inliers_every;
        if (inliers_dst && inliers_capacity < cap) return set_err(c, WASS_ERR_INVALID_ARG, "inliers_dst must hold %zu points", cap);
        const unsigned nb2 = nblk(cap);
        const size_t text_off = (256 + cap * 24 + (size_t)nb2 * 4 + 255) & ~(size_t)255;
        if (inliers_text_dst && inliers_text_capacity < cap * INL_LINE_MAX)
            return set_err(c, WASS_ERR_INVALID_ARG, "inliers_text_dst must hold %zu bytes", cap * (size_t)INL_LINE_MAX);
        if ((rc = ensure(c, inlbuf, text_off + (inliers_text_dst ? cap * INL_LINE_MAX + (size_t)nb2 * 256 * INL_LINE_MAX + 256 : 0)))) return rc;   // text, staging
        inl_total = (unsigned int*)inlbuf.p;                       // [0]: number of refinement inliers, [2]: bytes of text, [3]: numbers not formatted; points from byte 256
        unsigned int* bc = (unsigned int*)((char*)c->scratch.p + 64);
        double* dout = (double*)((char*)inlbuf.p + 256);
        WASS_HIP(c, hipStreamWaitEvent(s, fs.ev_copy, 0));        // the download (and the text kernels) that read THIS buffer: two frames ago
        hipLaunchKernelGGL(k_inlier_counts, dim3(nb), dim3(256), 0, s, m->valid, m->x, m->y, m->z, m->w, n, rd, bc);
        hipLaunchKernelGGL(k_scan_blocks, dim3(1), dim3(256), 0, s, bc, (int)nb, inl_total);
        hipLaunchKernelGGL(k_inlier_pack, dim3(nb), dim3(256), 0, s, m->valid, m->x, m->y, m->z, m->w, n, rd, (const unsigned int*)bc,
                           (unsigned)inliers_every, dout);
        inl_copy = inliers_dst ? cap * 24 : 0;                    // (text only: the points stay on the device, wass_ctx_frame_inliers fetches them on demand)
        if (inliers_text_dst) {                                   // the file's text, formatted here (fmt_g6.h)
            unsigned int* bb = (unsigned int*)((char*)inlbuf.p + 256 + cap * 24);
            char* staging = (char*)inlbuf.p + text_off + cap * INL_LINE_MAX;
            // (On the tail stream.  Putting these kernels on the copy stream behind an event, or into a small persistent grid held to one
            // wave per SIMD, changed nothing / made it worse: what they cost the C++ driver is their share of a tail stream that is
            // nearly as long as the SGM stage by now -- NOTES/tail_and_host.md, round 5.)
            WASS_HIP(c, hipMemsetAsync(inl_total + 2, 0, 8, s));
            hipLaunchKernelGGL(k_inl_text_format, dim3(nb2), dim3(256), 0, s, (const double*)dout, (const unsigned int*)inl_total, (unsigned)inliers_every, staging,
                               bb, inl_total + 3);
            hipLaunchKernelGGL(k_scan_blocks, dim3(1), dim3(256), 0, s, bb, (int)nb2, inl_total + 2);
            // Pinned host memory is mapped into the device's address space: the packing kernel then writes the text straight into the
            // caller's buffer -- exactly the file's bytes cross PCIe, where a copy would have to move the 40-bytes-per-point worst
            // case (the length is only known on the device).
            char* direct = nullptr;
            {
                hipPointerAttribute_t at;
                if (hipPointerGetAttributes(&at, inliers_text_dst) == hipSuccess && at.type == hipMemoryTypeHost && at.devicePointer) direct = (char*)at.devicePointer;
                else (void)hipGetLastError();
            }
            hipLaunchKernelGGL(k_inl_text_pack, dim3(nb2), dim3(256), 0, s, (const char*)staging, (const unsigned int*)bb, (const unsigned int*)(inl_total + 2), nb2,
                               direct ? direct : (char*)inlbuf.p + text_off);
            text_totals = inl_total + 2;
            if (direct) goto text_done;
            text_copy = cap * INL_LINE_MAX;
            text_src = (const char*)inlbuf.p + text_off;
        text_done:;
        }
        fs.inl_cap = cap;
    }
    hipLaunchKernelGGL(k_frame_rt, dim3(1), dim3(64), 0, s, ds);
    unsigned char* stage = nullptr;
    if ((rc = host_stage(c, &stage))) return rc;
    if ((rc = ensure(c, c->limits, NSLOT * 6 * 8))) return rc;
    unsigned long long* lim = (unsigned long long*)c->limits.p;            // [NSLOT][6] keys
    WASS_HIP(c, hipMemcpyAsync(lim, stage + 4096, NSLOT * 6 * 8, hipMemcpyHostToDevice, s));
    unsigned int* total = (unsigned int*)c->scratch.p;
    unsigned int* bcnt = (unsigned int*)((char*)c->scratch.p + 64);
    unsigned char* img = (unsigned char*)c->xyzc.p;
    WASS_HIP(c, hipStreamWaitEvent(s, ev_prev, 0));                       // the previous frame's download still reading the image
    hipLaunchKernelGGL(k_crop_limits_counts_dev, dim3((nb + CLC_CHUNKS - 1) / CLC_CHUNKS), dim3(256), 0, s, m->valid, m->x, m->y, m->z, n,
                       (const DevState*)ds, max_distance, kept1 + NSLOT, lim, bcnt, nb);
    hipLaunchKernelGGL(k_scan_blocks, dim3(1), dim3(256), 0, s, bcnt, (int)nb, total);
    hipLaunchKernelGGL(k_frame_header, dim3(1), dim3(64), 0, s, ds, (const unsigned long long*)lim, (const unsigned int*)total,
                       (const unsigned long long*)kept1, img, (const unsigned long long*)c->tri_cnt.p, (const unsigned int*)inl_total,
                       (const unsigned int*)nullptr);
    hipLaunchKernelGGL(k_xyzc_pack_dev, dim3(nb), dim3(256), 0, s, m->valid, m->x, m->y, m->z, n, (const DevState*)ds,
                       (const unsigned int*)bcnt, (uint16_t*)(img + 148));
    WASS_HIP(c, hipGetLastError());
    if (timed) { (void)hipEventRecord(c->ev_tail[5], s); c->tail_timed[c->tail_set] = true; }
    fs.tail_set = c->tail_set;
    if (text_totals) hipLaunchKernelGGL(k_inl_text_totals, dim3(1), dim3(64), 0, s, ds, text_totals);   // after the header kernel's write of the record
    WASS_HIP(c, hipEventRecord(c->ev_pack, s));
    WASS_HIP(c, hipStreamWaitEvent(c->copy, c->ev_pack, 0));
    WASS_HIP(c, hipMemcpyAsync(fs.h_frame, ds, sizeof(DevState), hipMemcpyDeviceToHost, c->copy));
    WASS_HIP(c, hipMemcpyAsync(dst, img, 148 + n * 6, hipMemcpyDeviceToHost, c->copy));
    if (inl_copy) WASS_HIP(c, hipMemcpyAsync(inliers_dst, (const char*)inlbuf.p + 256, inl_copy, hipMemcpyDeviceToHost, c->copy));
    // (the text's length is only known on the device: the copy takes what a frame of this size can need at most -- 0.1 ms per MB)
    if (text_copy) WASS_HIP(c, hipMemcpyAsync(inliers_text_dst, text_src, text_copy, hipMemcpyDeviceToHost, c->copy));
    if (component_mask_dst) WASS_HIP(c, hipMemcpyAsync(component_mask_dst, c->ccmask.p, n, hipMemcpyDeviceToHost, c->copy));
    WASS_HIP(c, hipEventRecord(fs.ev_copy, c->copy));
    c->ev_copy = fs.ev_copy;                  // "the most recently recorded download event"
    fs.inl_every = (inliers_dst || inliers_text_dst) ? inliers_every : 0;
    fs.inl_text = inliers_text_dst != nullptr;
    fs.sgm_call = c->nsgm;                    // the SGM call that fed this frame is the last one enqueued (0: none)
    c->frame_pending = true;
    ++c->nframe_enq;
    return WASS_OK;
}

int wass_format_g6(double v, char* out) { return out ? wass::fmt_g6(v, out) : -1; }

// the selected inlier points of the frame whose result was read last, fetched on demand (the text-only form of
// wass_mesh_finish_frame_async_ex2 leaves them on the device)
int wass_ctx_frame_inliers(wass_ctx* c, double* dst, size_t capacity_points, uint64_t* n_out)
{
    if (!c || !dst) return set_err(c, WASS_ERR_INVALID_ARG, "null argument");
    if (c->nframe_col == 0) return set_err(c, WASS_ERR_INVALID_ARG, "no collected frame");
    const int slot = (int)((c->nframe_col - 1) & 1);       // the frame whose record was read last; its buffer lives until the frame after next
    const wass_ctx::FrameSlot& fs = c->fslot[slot];
    const wass::Buf& inlbuf = slot ? c->inl2 : c->inl;
    if (fs.inl_every <= 0 || !inlbuf.p || !fs.h_frame) return set_err(c, WASS_ERR_INVALID_ARG, "no collected frame with an inlier selection");
    WASS_HIP(c, hipSetDevice(c->device));
    const DevState& h = *(const DevState*)fs.h_frame;
    const uint64_t n = ((uint64_t)h.ninl_sel + (uint64_t)fs.inl_every - 1) / (uint64_t)fs.inl_every;
    if (n > capacity_points || n > fs.inl_cap) return set_err(c, WASS_ERR_INVALID_ARG, "dst must hold %llu points", (unsigned long long)n);
    // (the frame's own downloads have completed -- its record was read -- so its selection is complete; a plain blocking copy)
    WASS_HIP(c, hipMemcpy(dst, (const char*)inlbuf.p + 256, (size_t)n * 24, hipMemcpyDeviceToHost));
    if (n_out) *n_out = n;
    return WASS_OK;
}

int wass_ctx_frame_result(wass_ctx* c, wass_frame_result* out)
{
    if (!c || !out) return set_err(c, WASS_ERR_INVALID_ARG, "null argument");
    if (c->nframe_col >= c->nframe_enq) return set_err(c, WASS_ERR_INVALID_ARG, "no wass_mesh_finish_frame_async call to wait for");
    const wass_ctx::FrameSlot& fs = c->fslot[c->nframe_col & 1];   // the OLDER of the pending frames
    WASS_HIP(c, hipEventSynchronize(fs.ev_copy));
    ++c->nframe_col;
    c->h_frame = fs.h_frame;
    const DevState& h = *(const DevState*)fs.h_frame;
    memset(out, 0, sizeof *out);
    out->zgap = h.zgap; out->n_gaps = h.sel_total; out->component_size = h.ccl_best >> 32;
    out->found = h.ransac_found; out->refine_ok = h.refine_ok;
    out->ransac_inliers = h.ransac_best;
    for (int k = 0; k < 4; ++k) { out->ransac_plane[k] = h.ransac_plane[k]; out->plane[k] = h.ransac_found && h.refine_ok ? h.plane[k] : NAN; }
    if (h.ransac_found) { out->kept_after_ransac_crop = h.kept1; out->kept_final = h.kept2; out->refine_inliers = (uint64_t)(h.ninl + 0.5); }
    out->n_points = h.npts;
    out->xyzc_bytes = 148 + (uint64_t)h.npts * 6;
    out->n_triangulated = h.ntri;
    if (c->tail_timed[fs.tail_set]) {                        // all six events lie before the download this function has waited for
        hipEvent_t* const ev = c->ev_tail_sets[fs.tail_set];       // the frame's own set: the next frame may have been triangulated already
        for (int k = 0; k < 5; ++k)
            if (hipEventElapsedTime(&out->stage_ms[k], ev[k], ev[k + 1]) != hipSuccess) out->stage_ms[k] = 0.0f;
    }
    out->n_inliers_out = fs.inl_every > 0 ? ((uint64_t)h.ninl_sel + (uint64_t)fs.inl_every - 1) / (uint64_t)fs.inl_every : 0;
    out->inliers_text_bytes = fs.inl_text ? h.inl_text_bytes : 0;
    out->inliers_text_unsupported = fs.inl_text ? h.inl_text_bad : 0;
    if (fs.sgm_call > 0 && c->nsgm - fs.sgm_call < (unsigned long long)wass_ctx::NSGM_SETS) {
        // status word of the frame's SGM call: copied to pinned memory in stream order long before the download this
        // function has just waited for; the slot is reused four calls later
        const uint32_t fl = c->h_flags[4 * (int)((fs.sgm_call - 1) % wass_ctx::NSGM_SETS)];
        out->sgm_cost_overflow = (int)(fl & 1);
        out->sgm_timeout = (int)((fl >> 1) & 1);
    } else {
        out->sgm_cost_overflow = out->sgm_timeout = -1;      // unknown: no SGM call of this context fed the frame
    }
    if (h.sel_fail == 2) return set_err(c, WASS_ERR_DEVICE, "radix select lost its rank (internal error)");
    return WASS_OK;
}

void wass_RT_from_plane(const double plane[4], double R[9], double T[3], double Rinv[9], double Tinv[3])
{
    rt_from_plane(plane, R, T, Rinv, Tinv);
}

int wass_mesh_encode_xyzc(wass_ctx* c, wass_mesh* m, const double plane[4], void** bytes, size_t* nbytes)
{
    if (!c || !m || !bytes || !nbytes) return set_err(c, WASS_ERR_INVALID_ARG, "null argument");
    const size_t cap = 148 + m->n() * 6;
    void* buf = malloc(cap);
    if (!buf) return set_err(c, WASS_ERR_NO_MEMORY, "out of host memory");
    int rc = wass_mesh_encode_xyzc_to(c, m, plane, buf, cap, nbytes);
    if (rc) { free(buf); return rc; }
    *bytes = buf;
    return WASS_OK;
}

static int encode_xyzc_impl(wass_ctx* c, wass_mesh* m, const double plane[4], void* dst, size_t capacity, size_t* nbytes, bool async)
{
    if (!c || !m || !dst || !nbytes) return set_err(c, WASS_ERR_INVALID_ARG, "null argument");
    WASS_HIP(c, hipSetDevice(c->device));
    double R[9] = { 1, 0, 0, 0, 1, 0, 0, 0, 1 }, T[3] = { 0, 0, 0 }, Rinv[9] = { 1, 0, 0, 0, 1, 0, 0, 0, 1 }, Tinv[3] = { 0, 0, 0 };
    if (plane) wass_RT_from_plane(plane, R, T, Rinv, Tinv);
    RTDev rt;
    memcpy(rt.R, R, sizeof R); memcpy(rt.T, T, sizeof T);
    const size_t n = m->n();
    const unsigned nb = nblk(n);
    int rc = ensure(c, c->scratch, 64 + (size_t)nb * 4 + 16);
    if (rc) return rc;
    if ((rc = ensure(c, c->counters, (size_t)NSLOT * 6 * 8))) return rc;
    // the packed triples get their own buffer: the download may still be in flight (async form) while the next
    // frame's stages reuse the scratch area
    if ((rc = ensure(c, c->xyzc, n * 6 + 16))) return rc;
    unsigned long long* lim = (unsigned long long*)c->counters.p;          // [NSLOT][6] keys
    unsigned int* total = (unsigned int*)c->scratch.p;
    unsigned int* bcnt = (unsigned int*)((char*)c->scratch.p + 64);
    uint16_t* dq = (uint16_t*)c->xyzc.p;
    unsigned long long init[NSLOT * 6];
    for (int i = 0; i < NSLOT; ++i) for (int k = 0; k < 6; ++k) init[i * 6 + k] = k < 3 ? ~0ull : 0ull;
    WASS_HIP(c, hipMemcpyAsync(lim, init, sizeof init, hipMemcpyHostToDevice, c->ts()));
    hipLaunchKernelGGL(k_xyzc_limits, dim3(1024), dim3(256), 0, c->ts(), m->valid, m->x, m->y, m->z, n, rt, lim);
    hipLaunchKernelGGL(k_block_counts, dim3(nb), dim3(256), 0, c->ts(), m->valid, n, bcnt);
    hipLaunchKernelGGL(k_scan_blocks, dim3(1), dim3(256), 0, c->ts(), bcnt, (int)nb, total);
    unsigned long long hs[NSLOT * 6], hl[6] = { ~0ull, ~0ull, ~0ull, 0, 0, 0 };
    unsigned int npts = 0;
    WASS_HIP(c, hipMemcpyAsync(hs, lim, sizeof hs, hipMemcpyDeviceToHost, c->ts()));
    WASS_HIP(c, hipMemcpyAsync(&npts, total, 4, hipMemcpyDeviceToHost, c->ts()));
    WASS_HIP(c, hipStreamSynchronize(c->ts()));
    for (int i = 0; i < NSLOT; ++i)
        for (int k = 0; k < 3; ++k) {
            if (hs[i * 6 + k] < hl[k]) hl[k] = hs[i * 6 + k];
            if (hs[i * 6 + 3 + k] > hl[3 + k]) hl[3 + k] = hs[i * 6 + 3 + k];
        }
    double mn[3], mx[3], sc[3];
    for (int k = 0; k < 3; ++k) {
        // no valid point: the reference writes +-DBL_MAX limits; keep that
        mn[k] = npts ? dunkey(hl[k]) : 1.7976931348623157e308;
        mx[k] = npts ? dunkey(hl[3 + k]) : -1.7976931348623157e308;
        sc[k] = 65535.0 / (mx[k] - mn[k]);
    }
    const size_t total_bytes = 148 + (size_t)npts * 6;
    if (total_bytes > capacity)
        return set_err(c, WASS_ERR_INVALID_ARG, "xyzC needs %zu bytes, buffer has %zu", total_bytes, capacity);
    if (npts) {
        WASS_HIP(c, hipStreamWaitEvent(c->ts(), c->ev_copy, 0));            // a previous download still reading dq
        hipLaunchKernelGGL(k_xyzc_pack, dim3(nb), dim3(256), 0, c->ts(), m->valid, m->x, m->y, m->z, n, rt, mn[0], mn[1], mn[2],
                           sc[0], sc[1], sc[2], (const unsigned int*)bcnt, dq);
    }
    unsigned char* buf = (unsigned char*)dst;
    size_t o = 0;
    const uint32_t n32 = npts;
    memcpy(buf + o, &n32, 4); o += 4;
    memcpy(buf + o, sc, 24); o += 24;
    memcpy(buf + o, mn, 24); o += 24;
    memcpy(buf + o, Rinv, 72); o += 72;
    memcpy(buf + o, Tinv, 24); o += 24;
    if (npts) {
        // the download runs on its own stream (a DMA engine), so the next frame's kernels do not wait for PCIe
        WASS_HIP(c, hipEventRecord(c->ev_pack, c->ts()));
        WASS_HIP(c, hipStreamWaitEvent(c->copy, c->ev_pack, 0));
        WASS_HIP(c, hipMemcpyAsync(buf + o, dq, (size_t)npts * 6, hipMemcpyDeviceToHost, c->copy));
        WASS_HIP(c, hipEventRecord(c->ev_copy, c->copy));
        if (!async) WASS_HIP(c, hipStreamSynchronize(c->copy));
    }
    *nbytes = total_bytes;
    return WASS_OK;
}

int wass_mesh_encode_xyzc_to(wass_ctx* c, wass_mesh* m, const double plane[4], void* dst, size_t capacity, size_t* nbytes)
{
    return encode_xyzc_impl(c, m, plane, dst, capacity, nbytes, false);
}

int wass_mesh_encode_xyzc_async(wass_ctx* c, wass_mesh* m, const double plane[4], void* dst, size_t capacity, size_t* nbytes)
{
    return encode_xyzc_impl(c, m, plane, dst, capacity, nbytes, true);
}

// np.nanmean over planes.txt rows (wassgridsurface.py:672-678): a row counts if none of its entries is NaN
// ("nan nan nan nan" is the only way wass_stereo writes NaN, wass_stereo.cpp:2104-2106)
void wass_planes_mean_accumulate(const double* planes, int n, double acc5[5])
{
    for (int i = 0; i < n; ++i) {
        const double* p = planes + (size_t)i * 4;
        if (p[0] != p[0] || p[1] != p[1] || p[2] != p[2] || p[3] != p[3]) continue;
        acc5[0] += p[0]; acc5[1] += p[1]; acc5[2] += p[2]; acc5[3] += p[3]; acc5[4] += 1.0;
    }
}

void wass_planes_mean_finish(const double acc5[5], double mean_out[4], int* n_valid)
{
    const double n = acc5[4];
    for (int k = 0; k < 4; ++k) mean_out[k] = n > 0 ? acc5[k] / n : NAN;
    if (n_valid) *n_valid = (int)(n + 0.5);
}

}  // extern "C"
