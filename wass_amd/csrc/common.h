// common.h -- shared declarations for libwassgpu (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <stdio.h>
#include <string.h>
#include <string>

#include "../../include/wass_gpu.h"

namespace wass {

// ---------------------------------------------------------------- packed u16
typedef unsigned short us2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ us2 pk_min(us2 a, us2 b) { return __builtin_elementwise_min(a, b); }
__device__ __forceinline__ us2 pk_max(us2 a, us2 b) { return __builtin_elementwise_max(a, b); }
__device__ __forceinline__ us2 pk_adds(us2 a, us2 b) { return __builtin_elementwise_add_sat(a, b); }
__device__ __forceinline__ us2 pk_subs(us2 a, us2 b) { return __builtin_elementwise_sub_sat(a, b); }
__device__ __forceinline__ us2 pk_splat(unsigned v) { us2 r; r.x = (unsigned short)v; r.y = (unsigned short)v; return r; }
__device__ __forceinline__ uint32_t as_u32(us2 v) { return __builtin_bit_cast(uint32_t, v); }
__device__ __forceinline__ us2 as_us2(uint32_t v) { return __builtin_bit_cast(us2, v); }

// lanes: DPP controls (gfx9 encoding)
enum : int {
    DPP_QUAD_XOR1 = 0xB1,       // quad_perm:[1,0,3,2]
    DPP_QUAD_XOR2 = 0x4E,       // quad_perm:[2,3,0,1]
    DPP_ROW_HALF_MIRROR = 0x141,
    DPP_ROW_MIRROR = 0x140,
    DPP_ROW_BCAST15 = 0x142,
    DPP_ROW_BCAST31 = 0x143,
    DPP_WAVE_SHL1 = 0x130,      // lane i <- lane i+1
    DPP_WAVE_SHR1 = 0x138,      // lane i <- lane i-1
};

template <int CTRL, int ROW_MASK = 0xf, int BANK_MASK = 0xf>
__device__ __forceinline__ uint32_t dpp_mov(uint32_t old, uint32_t src)
{
    return (uint32_t)__builtin_amdgcn_update_dpp((int)old, (int)src, CTRL, ROW_MASK, BANK_MASK, false);
}

// minimum of a u32 over the 64 lanes of the wave, returned uniformly (SGPR).
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v)
{
    v = min(v, dpp_mov<DPP_QUAD_XOR1>(0xFFFFFFFFu, v));
    v = min(v, dpp_mov<DPP_QUAD_XOR2>(0xFFFFFFFFu, v));
    v = min(v, dpp_mov<DPP_ROW_HALF_MIRROR>(0xFFFFFFFFu, v));
    v = min(v, dpp_mov<DPP_ROW_MIRROR>(0xFFFFFFFFu, v));          // every row of 16 uniform
    v = min(v, dpp_mov<DPP_ROW_BCAST15, 0xa>(0xFFFFFFFFu, v));    // rows 1,3 <- min(row, previous row)
    v = min(v, dpp_mov<DPP_ROW_BCAST31, 0xc>(0xFFFFFFFFu, v));    // rows 2,3 <- min(row, lane 31)
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

// two independent reductions in lock-step: each fills the other's DPP wait states
__device__ __forceinline__ void wave_min2_u32(uint32_t& a, uint32_t& b)
{
    a = min(a, dpp_mov<DPP_QUAD_XOR1>(0xFFFFFFFFu, a));
    b = min(b, dpp_mov<DPP_QUAD_XOR1>(0xFFFFFFFFu, b));
    a = min(a, dpp_mov<DPP_QUAD_XOR2>(0xFFFFFFFFu, a));
    b = min(b, dpp_mov<DPP_QUAD_XOR2>(0xFFFFFFFFu, b));
    a = min(a, dpp_mov<DPP_ROW_HALF_MIRROR>(0xFFFFFFFFu, a));
    b = min(b, dpp_mov<DPP_ROW_HALF_MIRROR>(0xFFFFFFFFu, b));
    a = min(a, dpp_mov<DPP_ROW_MIRROR>(0xFFFFFFFFu, a));
    b = min(b, dpp_mov<DPP_ROW_MIRROR>(0xFFFFFFFFu, b));
    a = min(a, dpp_mov<DPP_ROW_BCAST15, 0xa>(0xFFFFFFFFu, a));
    b = min(b, dpp_mov<DPP_ROW_BCAST15, 0xa>(0xFFFFFFFFu, b));
    a = min(a, dpp_mov<DPP_ROW_BCAST31, 0xc>(0xFFFFFFFFu, a));
    b = min(b, dpp_mov<DPP_ROW_BCAST31, 0xc>(0xFFFFFFFFu, b));
    a = (uint32_t)__builtin_amdgcn_readlane((int)a, 63);
    b = (uint32_t)__builtin_amdgcn_readlane((int)b, 63);
}

// ------------------------------------------------------------------ geometry
// Derived sizes of one padded SGBM problem.
struct SgmDims {
    int w, h;            // unpadded crop
    int Wp;              // padded width  = w + D + max(off,0)
    int D;               // numDisparities
    int NP;              // packed u16 pairs per lane: ceil(D / 128)
    int Dp;              // padded disparity count = 128 * NP
    int minD, maxD;      // minDisparity, minD + D
    int minX1;           // = maxD
    int width1;          // = Wp - maxD
    int SW2;             // blockSize / 2
    int P1, P2;
    int uniq, d12;
    int ftzero;
    int ndirs;
    int off_pos, comp;   // max(off,0), max(-off,0)
    int speckle_win = 0, speckle_range = 0;   // cv::filterSpeckles inside compute() when speckle_win > 0
    size_t cells() const { return (size_t)h * width1 * Dp; }
};

// pitch (in u16) of one mirrored image-2 plane; the slack absorbs reads of padded disparity slots past the row
constexpr int BT2_FRONT = 128;    // u16 elements of slack in front of the first plane (group loads may start a few elements early)
static inline int bt2_pitch(int Wp) { return (Wp + 1024 + 63) & ~63; }

struct Buf {
    void* p = nullptr;
    size_t cap = 0;
};

}  // namespace wass

struct wass_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err;
    // scratch HBM (grown on demand, never shrunk)
    wass::Buf bt1, bt2;            // BT interval records: bt1 8 B/pixel; bt2 six mirrored u16 planes per row
    wass::Buf hsum, C, S;          // u16 volumes [h][width1][Dp]
    wass::Buf ckpt;                // forward-path checkpoints of k_pair (1/K of a volume)
    wass::Buf sel_d16, sel_key;    // per (y,x): raw fixed-point disparity / (minS<<16|d)
    wass::Buf raw;                 // padded-width raw disparity [h][Wp] int16
    wass::Buf flags;               // u32[4]: [0] = cost overflow
    uint32_t* h_flags = nullptr;   // pinned host copy of flags[0], refreshed at the end of every SGM call (stream-ordered)
    wass::Buf tmp_in0, tmp_in1, tmp_out;   // staging for the host-pointer entry points
    wass::Buf tmp_mask;
    wass::Buf fA, fB, fC;          // float32 maps of the disparity clean-up
    wass::Buf fD, fE;              // DENSE_SCALE != 1: nearest / cubic copies at the output size (post.hip)
    wass::Buf uf;                  // union-find parent / size arrays of the optional component filters (post_opt.hip)
    wass::Buf rs_r, rs_l;          // DENSE_SCALE != 1: resized SGBM inputs
    wass::Buf raw2;                // speckle filter: median-filtered padded disparity
    wass::Buf grid;                // wass_mesh_grid_idw: accumulators and maps of the surface grid (grid.hip)
    wass::Buf counters;            // striped atomics of the mesh stages
    wass::Buf tri_cnt;             // striped count of the points the last triangulation produced (read by the frame tail)
    wass::Buf inl;                 // every n-th refinement inlier of the frame tail (plane_refinement_inliers.xyz)
    wass::Buf dstate;              // device-resident scalar record + radix histogram (mesh.hip DevState)
    wass::Buf scratch;             // mesh stages: gaps / labels / partial sums / packed output
    wass::Buf xyzc;                // packed u16 triples of mesh_cam.xyzC (own buffer: downloaded asynchronously)
    wass::Buf limits;              // striped min/max keys of the frame tail
    // Up to TWO frames may be pending (round 5): a driver enqueues frame n's tail BEFORE it reads frame n-1's record, so that the tail
    // stream never waits for the previous frame's downloads plus a host round trip (with one record per context that chain, not the
    // SGM stage, set the C++ driver's frame period).  Everything a pending frame's record needs is per slot; slot = frame number & 1.
    struct FrameSlot {
        void* h_frame = nullptr;   // pinned: the device state record as downloaded
        hipEvent_t ev_copy = nullptr;   // the frame's downloads have finished
        int inl_every = 0;         // > 0: the frame also selected every n-th refinement inlier
        bool inl_text = false;     // ... and formatted plane_refinement_inliers.xyz on the device
        size_t inl_cap = 0;        // ... capacity (points) of the selection
        unsigned long long sgm_call = 0;   // 1-based index of the SGM call whose disparity the frame was built from
        int tail_set = 0;          // the timing-event set its tail was recorded into
    } fslot[2];
    unsigned long long nframe_enq = 0, nframe_col = 0;   // frame tails enqueued / records read
    int ds_slot = 0;               // which of the two device state records (and inlier buffers) the stage helpers use
    wass::Buf inl2;                // the odd frames' selected inliers (the even frames' live in `inl`)
    void* h_frame = nullptr;       // (alias of the last collected slot's record)
    bool frame_pending = false;
    void* h_stage = nullptr;       // pinned source images of the frame tail's small H2D copies (mesh.hip host_stage)
    hipEvent_t ev_stage = nullptr, ev_stage2 = nullptr;   // the RANSAC triplets' staging areas (two, alternated) have been copied to the device
    hipEvent_t ev_producer = nullptr; // wass_ctx_wait_for_stream
    hipEvent_t ev_dl = nullptr;        // wass_download_async: orders the copy stream after the SGM stream
    hipEvent_t ev_dl_tail = nullptr;   // ... and after the tail stream
    bool stage_uv_busy = false;
    hipStream_t copy = nullptr;    // D2H of the xyzC payload
    // Everything after the SGM call (disparity clean-up, triangulation, mesh stages: post.hip, mesh.hip) is enqueued on
    // ts(): the main stream, or -- with tail overlap on -- a second stream that waits for the last SGM call, so that
    // this string of small, latency-bound kernels runs underneath the next frame's bandwidth-bound SGM stage.
    hipStream_t tail = nullptr;
    hipEvent_t ev_post = nullptr;  // the clean-up on the tail stream has read the disparity of the last SGM call
    bool tail_overlap = false;
    hipStream_t ts() const { return tail_overlap ? tail : stream; }
    hipEvent_t ev_pack = nullptr, ev_copy = nullptr;
    // stage times of the frame tail (wass_frame_result.stage_ms): [0] before k_triangulate, [1] entry of the mesh tail, [2] z-gap
    // percentile found, [3] biggest component kept, [4] RANSAC plane picked, [5] file image packed.  Created on first use.
    // Two sets, alternated by wass_triangulate[_dev]: a pipelined driver enqueues frame n+1's triangulation BEFORE it reads frame
    // n's record, so one set would be re-recorded under the reader (every frame but the last of a sequence had all-zero rows).
    hipEvent_t ev_tail_sets[4][6] = {};    // (four: with two frames pending a record is read after two more triangulations)
    hipEvent_t* ev_tail = ev_tail_sets[0];  // set of the last triangulation
    int tail_set = 0;
    bool tail_timed[4] = {};
    wass::Buf ccmask;              // valid mask after the outlier removal, kept for graph_components.jpg when asked for
    // the debug pictures rendered and JPEG-coded on the device (jpeg.hip): coefficient / offset / bit-stream scratch, Huffman tables,
    // result words; four tickets in flight (event + header sizes + pinned result words per ticket)
    wass::Buf jpeg_scratch, jpeg_huff, jpeg_out, jpeg_info, jpeg_part;
    bool jpeg_tables_ready = false;
    hipEvent_t ev_dbg[4] = {};
    unsigned long long dbg_tickets = 0;
    uint32_t dbg_hdr[4][8] = {};
    uint32_t* h_dbg_info = nullptr;
    // wass_upload_async: uploads in flight on the copy stream, by destination; consumers wait for the matching event
    struct UploadSlot { const char* dst = nullptr; size_t n = 0; hipEvent_t ev = nullptr; bool pending = false, consumed = false; };
    UploadSlot uploads[8];
    int upload_next = 0;
    wass::Buf rect_tab;            // fixed-point interpolation tables of the rectification resamplers (rectify.hip)
    bool rect_tab_ready = false;
    wass::Buf rect_mx, rect_my;    // staging for host-pointer map uploads
    // cv::undistort's normalised coordinate tables depend on (K, w, h) only: kept per camera so that the per-frame call of a
    // sequence neither recomputes nor synchronises (wass_undistort_dev)
    struct UndCache { double K[9] = {}; int w = 0, h = 0; bool valid = false; wass::Buf xy; } und_cache[2];
    int und_next = 0;
    wass::Buf clahe_lut;           // per-tile look-up tables of wass_clahe_dev
    // stage events of the SGM call, two sets used alternately so that the timings of call n can be read after call
    // n+1 has been enqueued (a lagging reader never stalls the pipeline)
    // (four sets, not two, since round 5: a driver that runs two frames ahead reads call n's timings after call n+2 has been enqueued)
    static constexpr int NSGM_SETS = 4;
    hipEvent_t evs[NSGM_SETS][8] = {};
    hipEvent_t* ev = evs[0];       // set of the last call
    unsigned long long nsgm = 0;   // SGM calls so far
    int launches[NSGM_SETS] = {};
    hipStream_t side = nullptr;    // checkpoint sweeps run ahead here
    hipEvent_t ev_cost = nullptr, ev_ckpt[4] = {};
    void* coll_comm = nullptr;     // ncclComm_t of wass_coll_init (coll.hip); coll_buf: 64 doubles of HBM for the all-reduce
    void* coll_buf = nullptr;
    int coll_world = 0;
    wass::SgmDims last = {};
    bool have_last = false;
    bool debug = false;            // keep the finished S volume for wass_sgm_debug_fetch
    // wass_ctx_set_kernel_events: every launch of the cost stage and of the aggregation family bracketed by two hipEvents on its own
    // stream (one set, re-recorded by every SGM call; read with wass_sgm_kernel_times after a synchronisation).  Off by default: an
    // event between two kernels is a marker packet on the queue.
    bool kernel_events = false;
    struct KernelEv { const char* name = nullptr; hipEvent_t a = nullptr, b = nullptr; };
    KernelEv kev[24];
    int kev_n = 0;
    wass_sgm_timings timings = {};
    bool timings_valid = false;
};

// organised point cloud in HBM (PovMesh, wass_stereo/PovMesh.h:33-51, as structure of arrays)
struct wass_mesh {
    int w = 0, h = 0;
    uint8_t* valid = nullptr;
    double* x = nullptr;
    double* y = nullptr;
    double* z = nullptr;
    uint8_t* gray = nullptr;
    uint8_t* codes = nullptr;      // why triangulate kept or rejected a pixel (R0 / R1 debug pictures), WASS_CODE_*
    size_t bytes = 0;
    int device = 0;
    const void* owner = nullptr;   // the context whose stream orders every use of this allocation
    size_t n() const { return (size_t)w * h; }
};

namespace wass {

int set_err(wass_ctx* c, int code, const char* fmt, ...);
int ensure(wass_ctx* c, Buf& b, size_t bytes);

#define WASS_HIP(ctx, call)                                                                 \
    do {                                                                                    \
        hipError_t e__ = (call);                                                            \
        if (e__ != hipSuccess)                                                              \
            return wass::set_err((ctx), WASS_ERR_DEVICE, "%s failed: %s (%s:%d)", #call,    \
                                 hipGetErrorString(e__), __FILE__, __LINE__);               \
    } while (0)

// Checkpoint regions of the chain families, in launch order.  With 8 paths the column family comes first: its
// forward checkpoints are produced by the cost stage itself (k_vsum_col walks whole columns top-down, which is
// exactly that family's forward path), so its pair kernel can start the moment C is complete.
// steps per checkpoint segment (and unroll depth of the sweeps) for NP packed pairs per lane: bounded by k_pair, which
// keeps 2 * K * NP vectors in registers (the cost and S rings) and 2 * K * NP per wave in LDS (the hand-over slots).
// NP = 3, 4 (D = 384, 512): K = 8 since the end of round 3 -- 64 ring registers and 64 KiB of LDS per workgroup, i.e. two
// workgroups per CU instead of four, and still faster than K = 4 (config E, alternating runs: aggregation 21.9 -> 21.2 ms,
// cost stage 5.1 -> 4.8 ms): half the checkpoint traffic (126 -> 118 GB per frame) outweighs the occupancy.  K = 6: the same.
#ifndef WASS_K_SMALL
#define WASS_K_SMALL 8
#endif
#ifndef WASS_K_MID
#define WASS_K_MID 8
#endif
// NP = 5 .. 8 (D = 640 .. 1024): K = 4 (2 until the end of round 3): at D = 1024 the hand-over slots of a workgroup are then
// the 64 KiB a kernel may declare; 2456 x 2058, 8-path: aggregation 35.2 -> 30.4 ms at D = 1024, 33.8 -> 28.8 ms at D = 768.
#ifndef WASS_K_BIG
#define WASS_K_BIG 4
#endif
constexpr int ckpt_k(int NP) { return NP <= 2 ? WASS_K_SMALL : (NP <= 4 ? WASS_K_MID : WASS_K_BIG); }

struct CkptLayout {
    int K = 8;                       // steps per checkpoint segment
    int nfam = 0;
    int dx[4] = {}, dy[4] = {}, smode[4] = {}, nch[4] = {}, mseg[4] = {};
    bool split[4] = {};              // family split in the middle (half_chain_geometry): nch counts sub-chains
    size_t off[5] = {};              // byte offsets into c->ckpt
    size_t moff[4] = {};             // byte offsets of the per-step minima records (K u16 per segment) into c->ckpt
    size_t total = 0;                // bytes of c->ckpt
    bool cols_from_cost = false;     // family 0 is the column family and its checkpoints come from k_vsum_col
    bool rows_fused = false;         // D <= 512: the row family is folded into the column family's pair kernel (k_pairx); 5 paths: path 2 + rows
    int nbx = 0;                     // ... blocks of 8 columns per row
    size_t roff[4] = {};             // ... byte offsets into c->ckpt: entry states of paths 0 / 4, minima of paths 0 / 4
    bool path2_from_cost = false;    // 5-path mode: k_vsum_col has already written S = L_2 (path 2 has no partner)
};
CkptLayout ckpt_layout(const SgmDims& d);

// brackets a launch with two events when the context asks for per-kernel times (wass_ctx_set_kernel_events); nothing otherwise
struct KernelClock {
    wass_ctx* c;
    explicit KernelClock(wass_ctx* ctx) : c(ctx) {}
    void begin(const char* name, hipStream_t s)
    {
        if (!c->kernel_events || c->kev_n >= (int)(sizeof c->kev / sizeof c->kev[0])) return;
        wass_ctx::KernelEv& e = c->kev[c->kev_n];
        if (!e.a && (hipEventCreate(&e.a) != hipSuccess || hipEventCreate(&e.b) != hipSuccess)) return;
        e.name = name;
        (void)hipEventRecord(e.a, s);
    }
    void end(hipStream_t s)
    {
        if (!c->kernel_events || c->kev_n >= (int)(sizeof c->kev / sizeof c->kev[0]) || !c->kev[c->kev_n].b) return;
        (void)hipEventRecord(c->kev[c->kev_n].b, s);
        ++c->kev_n;
    }
};

void coll_release(wass_ctx* c);               // coll.hip
// post_opt.hip: optional parts of sgbm_dense_stereo (row a9)
int speckle_filter_dev(wass_ctx* c, int16_t* img, int w, int h, int newVal, int maxSize, int maxDiff, hipStream_t s);
int biggest_component_dev(wass_ctx* c, float* disp, int w, int h, int threshold, uint8_t* flag, hipStream_t s);
int resize_inputs_dev(wass_ctx* c, const uint8_t* src, int w, int h, size_t pitch, uint8_t* dst, int ws, int hs, double fx, double fy,
                      hipStream_t s);
int resize_f32_dev(wass_ctx* c, const float* src, int sw, int sh, float* dst, int dw, int dh, bool cubic, hipStream_t s);
void mesh_pool_ctx_alive(const void* ctx, bool alive);   // mesh.hip: only live contexts get allocations parked for them
void mesh_pool_purge(const void* owner);      // mesh.hip: parked mesh allocations of a context that is going away

// stage launchers (each enqueues on c->stream)
int launch_prefilter(wass_ctx* c, const SgmDims& d, const uint8_t* d_img1, const uint8_t* d_img2, size_t pitch);
int launch_cost_volume(wass_ctx* c, const SgmDims& d);
int launch_vsum_only(wass_ctx* c, const SgmDims& d, bool plain);   // the vertical block sum alone, on the hsum volume of the last frame
int launch_aggregate(wass_ctx* c, const SgmDims& d, int* n_launches);
// wass_sgm_selftest: S of the last call (kept: debug mode) against one plain sweep per path into S2; mismatching cells are added to *d_count
int selftest_reference(wass_ctx* c, const SgmDims& d, uint32_t* S2, unsigned long long* d_count);
int launch_select(wass_ctx* c, const SgmDims& d);
int launch_median_crop(wass_ctx* c, const SgmDims& d, int16_t* d_out);
int wait_uploads(wass_ctx* c, const void* p, hipStream_t s);   // order s after the pending uploads that cover p
int launch_median_full(wass_ctx* c, const SgmDims& d, int16_t* d_padded_out);   // the whole padded map (speckle filter path)

}  // namespace wass
