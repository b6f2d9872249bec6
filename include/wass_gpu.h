/*
 * wass_gpu.h -- C ABI of libwassgpu.so: the MI355X (gfx950) implementation of
 * the wass_stereo dense-stereo hot path.
 *
 * The reference has no in-process plugin API for this path; wass_stereo is one
 * process whose main() (src/wass_stereo/wass_stereo.cpp:1799-2149) calls
 * sgbm_dense_stereo / triangulate / PovMesh methods directly.  Each entry
 * point below replaces one of those reference interfaces (cited file:line,
 * relative to /root/reference/src) so that a maintainer can swap the body of
 * the corresponding function for one call.
 *
 * Conventions
 *   - every function returns 0 (WASS_OK) or a negative wass_status; nothing
 *     throws or aborts; wass_last_error(ctx) gives a message.
 *   - plain pointers and sizes only.  "_dev" variants take DEVICE pointers and
 *     are asynchronous on the context's stream; the others take HOST pointers
 *     and return when the result is in host memory.
 *   - a context owns one GPU, one stream and its scratch HBM.  One host thread
 *     per context; distinct contexts are independent (one per GPU / process).
 */
#ifndef WASS_GPU_H
#define WASS_GPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    WASS_OK = 0,
    WASS_ERR_INVALID_ARG = -1,
    WASS_ERR_UNSUPPORTED = -2,   /* e.g. MAX_DISPARITY > 1024, WINSIZE > 17     */
    WASS_ERR_NO_MEMORY = -3,
    WASS_ERR_DEVICE = -4,        /* HIP runtime error, no GPU                  */
    WASS_ERR_COST_OVERFLOW = -5, /* int16 cost precondition violated (A.7)     */
    WASS_ERR_TOO_FEW_POINTS = -6
} wass_status;

typedef struct wass_ctx wass_ctx;

int wass_ctx_create(int device_id, wass_ctx** out);
/* Devices the HIP runtime of this process shows (0 without a usable GPU or runtime).  Lets a host tell "no GPU at all" (every entry
 * point fails loudly) from "this device index does not exist here" (another process, or device 0, can take the frame). */
int wass_device_count(int* n_devices);
void wass_ctx_destroy(wass_ctx* ctx);
const char* wass_last_error(const wass_ctx* ctx);
/* raw hipStream_t of the context (for callers that enqueue their own work) */
void* wass_ctx_stream(wass_ctx* ctx);
int wass_ctx_synchronize(wass_ctx* ctx);
/* The "_dev" entry points read caller-owned device buffers on the context's own (non-blocking) streams.  A caller
 * that produced those buffers on another stream calls this first: it orders all work enqueued on the context from now
 * on after everything already enqueued on producer_stream (a hipStream_t; NULL = the legacy default stream). */
int wass_ctx_wait_for_stream(wass_ctx* ctx, void* producer_stream);
/* Two-stage pipelining inside one context.  on != 0: every stage after the SGM call (wass_disparity_postprocess*,
 * wass_triangulate*, wass_mesh_*) is enqueued on a second stream that waits for the last wass_sgm_disparity_dev
 * call, so the tail of frame i (small, latency-bound kernels and the PCIe download) runs underneath the SGM stage
 * of frame i+1.  Reusing one disparity buffer for every frame is safe (the SGM call waits for the previous
 * frame's clean-up to have read it before its last kernel writes it); alternating two avoids that wait.
 * wass_ctx_synchronize waits for both streams. */
int wass_ctx_set_tail_overlap(wass_ctx* ctx, int on);
/* test mode: keep intermediates (the finished S volume) that production never writes to HBM */
int wass_ctx_set_debug(wass_ctx* ctx, int on);
const char* wass_version(void);

/* Asynchronous upload h_src -> d_dst on the context's copy stream (h_src: pinned host memory, valid until the copy has
 * executed).  The calls of this library that READ device inputs (wass_sgm_disparity_dev, wass_burned_area_mask_dev) wait
 * for the pending uploads that cover their input pointers, and everything later in a frame is ordered after the SGM
 * call; nothing else waits.  A driver that uploads frame i+1 just before it submits frame i (three input sets: the
 * buffer was last used by frame i-2) gets the transfer underneath frame i's kernels.  A kernel of the caller's own that
 * reads d_dst must be ordered by the caller (wass_ctx_synchronize, or pass the buffer through one of the calls above). */
int wass_upload_async(wass_ctx* ctx, void* d_dst, const void* h_src, size_t nbytes);
/* Device and pinned host memory for a host program that has no other GPU runtime of its own (the C++ sequence driver;
 * bench.py and the tests use torch's allocator instead): hipMalloc / hipHostMalloc on the context's device.  Device
 * memory comes back zero-filled.  Free with the matching call before the context is destroyed. */
int wass_device_alloc(wass_ctx* ctx, size_t nbytes, void** d_out);
void wass_device_free(wass_ctx* ctx, void* d_ptr);
/* d_src -> h_dst after everything enqueued on the context so far; returns when the bytes are in host memory */
int wass_download(wass_ctx* ctx, void* h_dst, const void* d_src, size_t nbytes);
/* cv::resize(src, dst, Size(dw, dh), 0, 0, INTER_CUBIC) of an 8-bit picture resident in HBM (the 0000000X_s.png previews of
 * load_data, wass_stereo.cpp:401-417): 11-bit fixed-point weights, the arithmetic of the DENSE_SCALE resampler.  On the context's
 * SGM stream; waits for pending uploads of d_src.  d_dst: dw * dh bytes, dense. */
int wass_resize_cubic_u8_dev(wass_ctx* ctx, const uint8_t* d_src, int sw, int sh, size_t src_stride, uint8_t* d_dst, int dw, int dh);
/* the same without waiting: the copy runs on the context's copy stream once everything enqueued so far on the SGM stream AND on
 * the tail stream (clean-up, triangulation, mesh stages under tail overlap) has finished; h_dst (pinned) is complete when a
 * later wass_ctx_frame_result() or wass_ctx_synchronize() returns */
int wass_download_async(wass_ctx* ctx, void* h_dst, const void* d_src, size_t nbytes);
int wass_pinned_alloc(wass_ctx* ctx, size_t nbytes, void** h_out);
void wass_pinned_free(wass_ctx* ctx, void* h_ptr);
/* DISCARD_BURNED_AREAS (wass_stereo.cpp:1072,1086): d_mask[i] = d_img[i] <= 254, on the context's SGM stream; feeds the
 * left_mask / right_mask arguments of wass_triangulate_dev.  Both pointers 4-byte aligned. */
int wass_burned_area_mask_dev(wass_ctx* ctx, const uint8_t* d_img, size_t n, uint8_t* d_mask);
/* The camera masks of triangulate() in general (wass_stereo.cpp:1057-1093): d_mask[i] = (d_file_mask ? d_file_mask[i] != 0 : 1)
 * && (d_img ? d_img[i] <= 254 : 1) -- LEFT_MASK_IMAGE / RIGHT_MASK_IMAGE thresholded by the caller (0/1 bytes), combined with
 * DISCARD_BURNED_AREAS on the device.  Either input may be NULL.  On the context's SGM stream; waits for pending uploads. */
int wass_camera_mask_dev(wass_ctx* ctx, const uint8_t* d_img, const uint8_t* d_file_mask, size_t n, uint8_t* d_mask);

/* ------------------------------------------------------------------------
 * cv::StereoSGBM parameters as sgbm_dense_stereo sets them
 * (wass_stereo/wass_stereo.cpp:742-759,772-782).
 *   ndirs = 5 : MODE_SGBM, what the reference runs (bit-exact parity mode)
 *   ndirs = 8 : MODE_HH  (the commented-out fullDP switch, :777)
 * ------------------------------------------------------------------------ */
typedef struct {
    int min_disp;         /* MIN_DISPARITY                                   */
    int num_disp;         /* MAX_DISPARITY, multiple of 16                   */
    int win;              /* WINSIZE (odd)                                   */
    int P1, P2;           /* DENSE_P1_MULT*win*win, DENSE_P2_MULT*win*win    */
    int uniq_ratio;       /* DENSE_UNIQUENESS_RATIO                          */
    int disp12_max_diff;  /* DENSE_DISP12MAXDIFF                             */
    int prefilter_cap;    /* DENSE_PREFILTER_CAP                             */
    int speckle_win;      /* DENSE_SPECKLE_WINDOW_SIZE (> 0: cv::filterSpeckles) */
    int speckle_range;    /* DENSE_SPECKLE_RANGE                             */
    int ndirs;            /* 5 or 8                                          */
    int disp_offset;      /* DISPARITY_OFFSET (:747,801-812)                 */
    double dense_scale;   /* DENSE_SCALE (:745)                              */
} wass_sgm_params;

/* Replaces wass_stereo.cpp:820-839: zero-pad both rectified crops, run
 * dense_stereo->compute(right_image, left_image, disparity), crop columns
 * [num_disp, num_disp + w).  right/left: w x h u8, pitch in bytes.
 * disp16_out: w x h int16 (4 fractional bits), in the right image's frame.
 * Returns WASS_ERR_COST_OVERFLOW (result still written) if a block cost
 * exceeded the int16 range the reference's scalar and SIMD builds agree on. */
/* DENSE_SCALE != 1 (:788-796): both crops are first resized with cv::resize INTER_CUBIC -- by (scale, 1) when the scale is
 * above 1, by (scale, scale) below -- and disp16_out has the size of the RESIZED crops: */
int wass_dense_input_size(int w, int h, double dense_scale, int* ws, int* hs);
int wass_sgm_disparity(wass_ctx* ctx, const uint8_t* right, const uint8_t* left,
                       int w, int h, size_t pitch, const wass_sgm_params* p,
                       int16_t* disp16_out);
int wass_sgm_disparity_dev(wass_ctx* ctx, const uint8_t* d_right, const uint8_t* d_left,
                           int w, int h, size_t pitch, const wass_sgm_params* p,
                           int16_t* d_disp16_out);

/* Stage timings of the last wass_sgm_disparity[_dev] call, measured with
 * hipEvents on the context's stream (milliseconds).  Synchronises. */
typedef struct {
    float prefilter_ms;   /* K1: Sobel/BT interval images                     */
    float cost_ms;        /* K2: block-summed cost volume C                   */
    float aggregate_ms;   /* K3: all path sweeps (the roofline kernel family) */
    float select_ms;      /* K4: WTA/uniqueness/subpixel/disp2/L-R            */
    float median_ms;      /* K5: median 3x3 + crop                            */
    float total_ms;
    int   aggregate_launches;
    int   cost_overflow;  /* 1 if the int16 precondition was violated         */
    float vsum_ms;        /* part of cost_ms: the vertical block sum, which also
                             runs path 2 (column checkpoints / S = L_2)       */
} wass_sgm_timings;
int wass_sgm_last_timings(wass_ctx* ctx, wass_sgm_timings* out);
/* the call before the last one: lets a pipelined driver read frame n's stage times after frame n+1 has been
 * enqueued, without waiting for frame n+1 */
int wass_sgm_prev_timings(wass_ctx* ctx, wass_sgm_timings* out);
/* number of wass_sgm_disparity[_dev] calls of this context that were enqueued completely (a failed call does not count): a
 * pipelined driver remembers the value after its frame's call and later picks last / prev timings by the difference */
int wass_sgm_call_count(wass_ctx* ctx, uint64_t* n_calls);
/* stage times of call number `call` (1-based, as counted by wass_sgm_call_count right after the call); the last four calls are kept */
int wass_sgm_call_timings(wass_ctx* ctx, uint64_t call, wass_sgm_timings* out);

/* Device-side canary for the aggregation kernels: runs one synthetic w x h pair with num_disp disparities through the
 * production schedule (checkpoint sweeps, pair kernels with recomputation, row fusion) and through one plain sweep per path,
 * and compares the two aggregated volumes S cell by cell ON THE DEVICE -- no CPU oracle, usable in the field.  *mismatches
 * receives the number of differing cells; returns WASS_ERR_DEVICE when it is not zero.  The two forms share the arithmetic
 * of one path step and nothing of the scheduling around it, which is where an instance-specific miscompile and a hardware
 * store hazard were found.  __graft_entry__.smoke() and tests/test_sgm_gpu.py run it for every NP. */
int wass_sgm_selftest(wass_ctx* ctx, int w, int h, int num_disp, int ndirs, uint64_t* mismatches);

/* Measurement hook for the roofline accounting (bench.py): re-runs the vertical block sum of the LAST call on its retained
 * horizontal sums, once as the plain sum (plain_ms) and once in the form the call used (production_ms; in 8-path mode it
 * also carries paths 2 / 6 and their checkpoints, in 5-path mode path 2 and S = L_2), best of three, hipEvents on the
 * context's stream.  production_ms - plain_ms is what the column paths add to the cost stage.  Synchronises. */
int wass_sgm_probe_vsum(wass_ctx* ctx, float* plain_ms, float* production_ms);

/* Measurement hook (bench.py, "kernel_ms"): with on != 0 every kernel launch of the cost stage and of the aggregation family is
 * bracketed by two hipEvents on the stream it is launched on (main or side).  wass_sgm_kernel_times then returns the LAST SGM call's
 * launches in launch order: their names, '\n'-separated, into names[names_cap], and their durations into ms[max_kernels]; *n_kernels
 * = how many.  Off by default (an event between two kernels is a marker packet on the queue); synchronises. */
int wass_ctx_set_kernel_events(wass_ctx* ctx, int on);
int wass_sgm_kernel_times(wass_ctx* ctx, char* names, size_t names_cap, float* ms, int max_kernels, int* n_kernels);

/* Test hooks: copy intermediates of the last wass_sgm_disparity call to host.
 * C/S are [h][width1][num_disp] int16 with width1 = w + max(disp_offset,0) -
 * min_disp (C without the +P2 bias); raw is the padded-width disparity before
 * the 3x3 median ([h][w + num_disp + max(disp_offset,0)]).  Any may be NULL.
 * S_out requires wass_ctx_set_debug(ctx, 1) before the disparity call. */
int wass_sgm_debug_fetch(wass_ctx* ctx, int16_t* C_out, int16_t* S_out, int16_t* raw_out);


/* ------------------------------------------------------------------------
 * Disparity clean-up, rows a7-a9.  Replaces wass_stereo.cpp:853-945 (this form:
 * DENSE_SCALE == 1, see _ex below): clean_and_convert_disparity (:714-733), DISP_DILATE_STEPS
 * x matrix_dilate_zero (:617-662, including its column-shift quirk),
 * DISP_EROSION_STEPS x matrix_erode_zero (:665-711), the same-size
 * NN/cubic resize + extra erosion mask (:903-928) and the optional
 * cv::medianBlur (:941-945; MEDIAN_FILTER_WSIZE 0, 3 or 5).
 * disp16: w x h int16 as produced by wass_sgm_disparity; out: w x h float32.
 * ------------------------------------------------------------------------ */
int wass_disparity_postprocess(wass_ctx* ctx, const int16_t* disp16, int w, int h,
                               const wass_sgm_params* p, int dilate_steps, int erode_steps,
                               int median_wsize, float* disp_f32_out);
int wass_disparity_postprocess_dev(wass_ctx* ctx, const int16_t* d_disp16, int w, int h,
                                   const wass_sgm_params* p, int dilate_steps, int erode_steps,
                                   int median_wsize, float* d_disp_f32_out);
/* The same with every option of wass_stereo.cpp:853-986: disp16 is ws x hs (wass_dense_input_size), the result is
 * out_w x out_h = roi_comb_right.size(): the converted map is multiplied by 1/DENSE_SCALE (:853), resized with
 * cv::resize INTER_NEAREST and INTER_CUBIC (:903-904), masked by the eroded nearest copy (:908-928), median-filtered,
 * and -- cc_threshold = DENSE_DISPARITY_BIGGEST_COMPONENT_THRESHOLD > 0 -- zeroed where the squared 3x3 Sobel gradient
 * exceeds the threshold and outside the largest 8-connected component of what is left (:947-986). */
int wass_disparity_postprocess_ex(wass_ctx* ctx, const int16_t* disp16, int ws, int hs, const wass_sgm_params* p,
                                  int dilate_steps, int erode_steps, int median_wsize, int cc_threshold,
                                  int out_w, int out_h, float* disp_f32_out);
int wass_disparity_postprocess_ex_dev(wass_ctx* ctx, const int16_t* d_disp16, int ws, int hs, const wass_sgm_params* p,
                                      int dilate_steps, int erode_steps, int median_wsize, int cc_threshold,
                                      int out_w, int out_h, float* d_disp_f32_out);
/* :947-986 alone, in place on a device map (test hook / building block) */
int wass_biggest_component_by_gradient_dev(wass_ctx* ctx, float* d_disp, int w, int h, int threshold);
/* the "large gradient" mask (non-zero where the squared Sobel magnitude exceeded the threshold, :951-957) of the last
 * component extraction on this context (cc_threshold > 0 in wass_disparity_postprocess_ex or the call above), w x h bytes to
 * the host: what the reference paints into disparity_large_gradient.jpg (:958-960) */
int wass_large_gradient_mask(wass_ctx* ctx, int w, int h, uint8_t* mask_out);

/* ------------------------------------------------------------------------
 * Triangulation, rows a10-a13.  Replaces triangulate(StereoMatchEnv&)
 * (wass_stereo.cpp:1039-1386), StereoMatchEnv::unrectify (:299-324) and
 * triangulate(p,q,R,T) (wass_lib/triangulate.hpp:26-72).  All matrices are
 * row-major doubles.
 * ------------------------------------------------------------------------ */
typedef struct {
    double K_left[9], K_right[9];   /* env.intrinsics_left / _right                 */
    double R[9], T[3];              /* env.R, env.T (|T| = 1, :360-370)             */
    int    use_custom;              /* USE_CUSTOM_STEREORECTIFY                     */
    double R1[9], R2[9];            /* env.rec_R1 / rec_R2        (OpenCV path)     */
    double P1[12], P2[12];          /* env.rec_P1 / rec_P2 (3x4)  (OpenCV path)     */
    double HLi[9], HRi[9];          /* env.HLi / HRi              (custom path)     */
    double disparity_compensation;  /* env.disparity_compensation (:804-812)        */
    double dense_scale;             /* DENSE_SCALE                                  */
} wass_geom;

typedef struct {
    double min_angle_deg;           /* TRIANG_MIN_ANGLE                             */
    double bbox[4];                 /* left, top, right, bottom in the left image;
                                       defaults (0,0,cols,rows) (:1046-1054)        */
    double cam_distance;            /* env.cam_distance (= 1)                       */
} wass_tri_params;

/* organised point cloud on the device: PovMesh (wass_stereo/PovMesh.h:28-88) as
 * structure-of-arrays (valid u8, x/y/z f64, gray u8), index v*width+u */
typedef struct wass_mesh wass_mesh;

/* disp_roi: roi_r[2] x roi_r[3] float32, the part of env.disparity inside
 * roi_comb_right (the reference map is zero elsewhere).  W,H: size of the
 * rectified frames.  roi_* = {x, y, width, height}.  right_img: the undistorted
 * RIGHT image (env.right, img_w x img_h) sampled for the point grey value
 * (:1342).  left_mask/right_mask: 0/1 images of the originals' size or NULL
 * (= all ones) (:1057-1093).  Host pointers; *_dev takes device pointers.
 * n_pts may be NULL for the *_dev form: the point count is then not read back
 * and the call does not synchronise with the host. */
int wass_triangulate(wass_ctx* ctx, const float* disp_roi, int W, int H,
                     const int roi_l[4], const int roi_r[4], const wass_geom* g,
                     const uint8_t* right_img, int img_w, int img_h,
                     const uint8_t* left_mask, const uint8_t* right_mask,
                     const wass_tri_params* tp, wass_mesh** out, uint64_t* n_pts);
int wass_triangulate_dev(wass_ctx* ctx, const float* d_disp_roi, int W, int H,
                         const int roi_l[4], const int roi_r[4], const wass_geom* g,
                         const uint8_t* d_right_img, int img_w, int img_h,
                         const uint8_t* d_left_mask, const uint8_t* d_right_mask,
                         const wass_tri_params* tp, wass_mesh** out, uint64_t* n_pts);
/* ---- the debug pictures of a frame, rendered and JPEG-coded on the device (replaces the cv::imwrite calls of wass_stereo.cpp:833, 854,
 * 1001, 1017, 1381-1382, 1925 and PovMesh.cpp:982-984).  Baseline JPEG, quality 95 like cv::imwrite's default,
 * 4:4:4, a restart marker after every row of blocks; the same bytes as the host writer of wass_amd/host/jpeg.hpp gives for the same
 * pixels (shared integer arithmetic, csrc/jpeg_spec.h). */
enum {
    WASS_PIC_STEREO = 0,           /* stereo.jpg                   rectified pair side by side, ROI rectangles, a red line every 20 rows */
    WASS_PIC_STEREO_INPUT = 1,     /* stereo_input.jpg             the two zero-padded SGBM inputs, left above right */
    WASS_PIC_DISPARITY_RAW = 2,    /* disparity_stereo_ouput.jpg   render_disparity_float of the converted raw disparity */
    WASS_PIC_DISPARITY_FINAL = 3,  /* disparity_final_scaled.jpg   ... of the final map */
    WASS_PIC_COVERAGE = 4,         /* disparity_coverage.jpg       right picture, green where disparity > 1, half size */
    WASS_PIC_R0 = 5,               /* undistorted/R0.jpg           grey where a point was triangulated, else the rejecting test's colour */
    WASS_PIC_R1 = 6,               /* undistorted/R1.jpg           the same with the matched left pixel's grey */
    WASS_PIC_COMPONENTS = 7,       /* graph_components.jpg         biggest component green, the other points blue, half size */
    WASS_DEBUG_PICTURES = 8
};
typedef struct wass_debug_desc {
    int W0, H0;                        /* size of the full rectified pictures */
    int roi_l[4], roi_r[4];            /* x, y, width, height of the two crops (equal sizes) */
    const uint8_t* d_left_crop;        /* the rectified crops the SGM stage was given (dense, roi-sized, in HBM) */
    const uint8_t* d_right_crop;
    const int16_t* d_disp16;           /* wass_sgm_disparity_dev's output for them */
    const float* d_dispf;              /* wass_disparity_postprocess_dev's output */
    int num_disp, min_disp, disp_offset;
    double disparity_compensation;
    int quality;                       /* 0 = 95 */
} wass_debug_desc;
/* One picture that exists in HBM (grey: channels 1, or r,g,b interleaved: 3) as a complete JPEG file in h_dst; synchronous. */
int wass_jpeg_encode_dev(wass_ctx* ctx, const uint8_t* d_pixels, int w, int h, int channels, size_t pitch_bytes, int quality,
                         uint8_t* h_dst, size_t capacity, size_t* nbytes);
/* width, height and channels (1 grey, 3 colour) of picture k for this frame geometry: what a caller sizes its slots by */
int wass_debug_picture_size(const wass_debug_desc* desc, int k, int* width, int* height, int* channels);
/* All eight pictures of the frame whose mesh is `mesh`, enqueued behind the frame's tail (call it after wass_mesh_finish_frame_async*
 * with a component mask destination, before destroying the mesh).  h_dst: pinned host memory (wass_pinned_alloc); picture k is written
 * as a complete file at h_dst + offset[k] by the kernels themselves, offset[0] = 0, offset[k + 1] = offset[k] + capacity[k] rounded up
 * to a multiple of 64.  Nothing is synchronised: wass_debug_pictures_result(ticket) waits for this frame's pictures and gives their
 * sizes (0: the picture did not fit into its slot and was not written).  Four tickets may be outstanding. */
int wass_debug_pictures_async(wass_ctx* ctx, const wass_mesh* mesh, const wass_debug_desc* desc, uint8_t* h_dst,
                              const size_t capacity[WASS_DEBUG_PICTURES], uint64_t* ticket);
int wass_debug_pictures_result(wass_ctx* ctx, uint64_t ticket, size_t nbytes[WASS_DEBUG_PICTURES]);

/* Why triangulate kept or rejected each pixel of the grid: what the reference paints into its debug pictures
 * undistorted/R0.jpg (low nibble) and R1.jpg (high nibble), wass_stereo.cpp:1111-1119,1216-1338.  codes_out: width x
 * height bytes (host). */
enum {
    WASS_CODE_NONE = 0,            /* not processed (no disparity): black */
    WASS_CODE_GREY = 1,            /* triangulated: the rectified image's grey value */
    WASS_CODE_OUTSIDE_IMAGE = 2,   /* teal   (255,255,0) BGR */
    WASS_CODE_OUTSIDE_BBOX = 3,    /* yellow (0,255,255): outside the bounding box or masked out */
    WASS_CODE_ANGLE = 4,           /* green  (0,255,0): rays too parallel */
    WASS_CODE_TOO_CLOSE = 5,       /* blue   (255,0,0) */
    WASS_CODE_TOO_DISTANT = 6      /* red    (0,0,255) */
};
int wass_mesh_reject_codes(wass_ctx* ctx, const wass_mesh* m, uint8_t* codes_out);
void wass_mesh_destroy(wass_mesh* m);
int wass_mesh_size(const wass_mesh* m, int* width, int* height);
/* copy the cloud to host (test hook, PLY / xyzbin writers): valid[w*h],
 * p3d[w*h*3] (interleaved xyz), gray[w*h]; any may be NULL */
int wass_mesh_download(wass_ctx* ctx, const wass_mesh* m, uint8_t* valid, double* p3d, uint8_t* gray);
/* build a mesh from host arrays (test hook) */
int wass_mesh_upload(wass_ctx* ctx, int width, int height, const uint8_t* valid, const double* p3d,
                     const uint8_t* gray, wass_mesh** out);

/* ------------------------------------------------------------------------
 * PovMesh stages, rows a14-a20 (wass_stereo/PovMesh.cpp).
 * ------------------------------------------------------------------------ */
/* compute_zgap_percentile (:888-926); exact order statistic; NaN if no gaps */
int wass_mesh_zgap_percentile(wass_ctx* ctx, wass_mesh* m, double percentile, double* out, uint64_t* n_gaps);
/* cluster_biggest_connected_component (:929-987): keep the largest 4-connected
 * component of valid points whose |dz| < zgap (first in column-major order on ties) */
int wass_mesh_keep_biggest_component(wass_ctx* ctx, wass_mesh* m, double zgap, uint64_t* size_out);
/* ransac_find_plane (:665-777).  uv_triplets[rounds][6] = {u1,v1,u2,v2,u3,v3}
 * drawn by the caller with rand() as in :680-691 (wass_ransac_sample does that).
 * Returns WASS_OK with *found = 0 when best < width*height/10 (:773). */
int wass_ransac_sample(int width, int height, int rounds, int32_t* uv_triplets);
/* The same draw from a PRIVATE generator that restates glibc's srand(seed) + rand(): what the reference gets for
 * RANDOM_SEED = seed (wass_stereo.cpp:1864-1872; RANSAC is its only rand() consumer), independent of whoever else in
 * the process calls rand() -- threads of the HIP runtime do. */
int wass_ransac_sample_seeded(uint32_t seed, int width, int height, int rounds, int32_t* uv_triplets);
int wass_mesh_ransac_plane(wass_ctx* ctx, wass_mesh* m, const int32_t* uv_triplets, int rounds,
                           double thr, double plane_out[4], uint64_t* best_inliers, int* found);
/* crop_plane (:780-815) */
int wass_mesh_crop_plane(wass_ctx* ctx, wass_mesh* m, const double plane[4], double thr, uint64_t* kept);
typedef struct {
    double xmin, xmax, ymin, ymax;  /* PLANE_REFINE_{XMIN,XMAX,YMIN,YMAX}           */
    double max_distance;            /* PLANE_REFINEMENT_MAX_DISTANCE                */
    int weight_by_distance;         /* PLANE_WEIGHT_PROPORTIONAL_TO_DISTANCE        */
    int central_third_only;         /* PLANE_USE_CENTRAL_THIRD_ONLY                 */
} wass_refine_params;
/* refine_plane (:581-660): weighted PCA plane through the inliers */
int wass_mesh_refine_plane(wass_ctx* ctx, wass_mesh* m, const wass_refine_params* rp,
                           double plane_out[4], uint64_t* n_inliers);
/* plane_refinement_inliers.xyz (wass_stereo.cpp:2077-2085): the points main() writes after refine_plane -- every
 * `every`-th (10) refinement inlier of PovMesh.cpp:590-606 in raster order -- selected on the device.  *xyz_out: malloc'ed
 * n_out x 3 doubles (release with wass_free); NULL when there is none. */
int wass_mesh_refinement_inliers(wass_ctx* ctx, wass_mesh* m, const wass_refine_params* rp, int every, double** xyz_out,
                                 uint64_t* n_out);
/* Fused forms of the calls above for throughput: same results, but every intermediate decision (radix-select
 * digits, winning component, best RANSAC candidate, centroid, 3x3 eigenvector) is taken on the device and the
 * host reads back once.
 *   wass_mesh_remove_outliers = zgap_percentile + keep_biggest_component   (wass_stereo.cpp:2046-2050)
 *   wass_mesh_fit_plane       = ransac_plane -> crop_plane(ransac_thr) -> refine_plane -> crop_plane(max_distance)
 *                               (wass_stereo.cpp:2062-2107); found == 0: nothing cropped, plane = NaN */
int wass_mesh_remove_outliers(wass_ctx* ctx, wass_mesh* m, double percentile, double* zgap_out, uint64_t* n_gaps,
                              uint64_t* size_out);
typedef struct {
    int      found;                 /* RANSAC succeeded (best >= width*height/10)   */
    double   ransac_plane[4];
    uint64_t ransac_inliers;
    double   plane[4];              /* refined plane                                 */
    uint64_t refine_inliers;
    uint64_t kept_after_ransac_crop, kept_final;
} wass_plane_result;
int wass_mesh_fit_plane(wass_ctx* ctx, wass_mesh* m, const int32_t* uv_triplets, int rounds, double ransac_thr,
                        const wass_refine_params* rp, double max_distance, wass_plane_result* out);
/* Everything main() does with the mesh after triangulation (wass_stereo.cpp:2046-2123: z-gap percentile, biggest
 * component, RANSAC plane, crop, refine, crop, mesh_cam.xyzC) enqueued WITHOUT a host synchronisation: all
 * decisions are taken on the device (RANSAC failure -> nothing cropped, identity R|T in the header, like
 * wass_mesh_encode_xyzc(plane = NULL)).  dst receives the file image (148-byte header + 6 bytes per point; it must
 * hold 148 + 6*width*height bytes, pinned memory recommended) through the context's copy stream.
 * wass_ctx_frame_result() waits for that download and reports what the stage-by-stage calls would have returned;
 * the number of valid bytes in dst is result.xyzc_bytes.  TWO frames may be pending per context (round 5): a driver enqueues
 * frame n+1's tail before it reads frame n's record, so that the tail stream goes from one frame's tail straight into the next
 * one's; wass_ctx_frame_result() hands the records out in submission order, a third wass_mesh_finish_frame_async* call without a
 * read in between is refused.  dst (and inliers_dst / inliers_text_dst) of a pending frame must stay untouched until its record
 * has been read. */
typedef struct {
    double   zgap;  uint64_t n_gaps, component_size;
    int      found, refine_ok;      /* refine_ok == 0 with found == 1: fewer than 3 refinement inliers (the
                                       stage-by-stage call returns WASS_ERR_TOO_FEW_POINTS)                     */
    double   ransac_plane[4];  uint64_t ransac_inliers;
    double   plane[4];         uint64_t refine_inliers, kept_after_ransac_crop, kept_final;
    uint64_t n_points, xyzc_bytes;
    /* status of the wass_sgm_disparity_dev call that produced the frame's disparity (the asynchronous SGM entry point
     * cannot return it): 1 = block cost + P2 left the int16 range where the reference is defined (the synchronous call
     * returns WASS_ERR_COST_OVERFLOW), -1 = unknown (the disparity did not come from this context's last two calls) */
    int      sgm_cost_overflow, sgm_timeout;
    /* valid points the last wass_triangulate[_dev] call of this context produced ("N valid points found",
     * wass_stereo.cpp:1374; the MIN_TRIANGULATED_POINTS test of :1993) -- counted on the device, no synchronisation */
    uint64_t n_triangulated;
    /* points written to inliers_dst by wass_mesh_finish_frame_async_ex (0 for the plain form or when RANSAC failed) */
    uint64_t n_inliers_out;
    /* GPU time (hipEvents on the context's tail stream, milliseconds) of the reference's timer rows after "Dense Stereo"
     * (wass_stereo.cpp:1982,2047,2049,2065,2089): [0] Triangulation, [1] Z-gap stats, [2] Outlier removal, [3] Plane fitting,
     * [4] Plane refinement (crop, refinement, crop, mesh_cam.xyzC image).  Zero when the mesh did not come from a
     * wass_triangulate[_dev] call of this context.  Under tail overlap the tail shares the GPU with the next frame's SGM stage. */
    float stage_ms[5];
    int reserved;
    /* wass_mesh_finish_frame_async_ex2: bytes of plane_refinement_inliers.xyz text in inliers_text_dst, and how many of its numbers
     * the device formatter could not write (inf, nan, |v| >= 1e6 or < 1e-22; never a triangulated coordinate): when that is not
     * zero the text is NOT the file -- format inliers_dst on the host instead */
    uint64_t inliers_text_bytes;
    uint32_t inliers_text_unsupported, reserved2;
} wass_frame_result;
int wass_mesh_finish_frame_async(wass_ctx* ctx, wass_mesh* m, double percentile, const int32_t* uv_triplets, int rounds,
                                 double ransac_thr, const wass_refine_params* rp, double max_distance, void* dst,
                                 size_t capacity);
/* The same, and additionally what main() writes to plane_refinement_inliers.xyz (wass_stereo.cpp:2077-2085): every
 * `inliers_every`-th refinement inlier (PovMesh.cpp:590-606) in raster order, selected on the device between the
 * refinement and the final crop and downloaded next to the file image -- inliers_dst: pinned host memory for
 * inliers_capacity points of 3 doubles (ceil(width*height / inliers_every) is always enough); NULL: not wanted.
 * component_mask_dst: pinned host memory for width*height bytes, the validity mask as cluster_biggest_connected_component leaves
 * it (PovMesh.cpp:929-987) -- before the plane stages crop it further; what graph_components.jpg is drawn from; NULL: not wanted. */
int wass_mesh_finish_frame_async_ex(wass_ctx* ctx, wass_mesh* m, double percentile, const int32_t* uv_triplets, int rounds,
                                    double ransac_thr, const wass_refine_params* rp, double max_distance, void* dst,
                                    size_t capacity, double* inliers_dst, size_t inliers_capacity, int inliers_every,
                                    uint8_t* component_mask_dst);
/* The same, and the file's TEXT as well: "x y z\n" per selected inlier, every number as a default-constructed std::ofstream
 * prints a double (precision 6, %g -- wass_stereo.cpp:2077-2085), formatted ON THE DEVICE (csrc/fmt_g6.h: correctly rounded in
 * 128-bit integer arithmetic, the characters of printf("%g")).  1.4 million numbers per 5-megapixel frame are a third of a
 * worker's host time when the host formats them.  inliers_text_dst: pinned host memory for 40 * ceil(width*height /
 * inliers_every) bytes (a line is at most 39 bytes); the valid length comes back in wass_frame_result.inliers_text_bytes.  When
 * the buffer is pinned host memory (wass_pinned_alloc, hipHostMalloc) the kernel writes the text straight into it and exactly the
 * file's bytes cross PCIe.  inliers_dst may be NULL then (the points are not downloaded); should inliers_text_unsupported come
 * back non-zero, wass_ctx_frame_inliers() fetches them for the host's formatter -- before the next frame's tail is enqueued. */
int wass_mesh_finish_frame_async_ex2(wass_ctx* ctx, wass_mesh* m, double percentile, const int32_t* uv_triplets, int rounds,
                                     double ransac_thr, const wass_refine_params* rp, double max_distance, void* dst,
                                     size_t capacity, double* inliers_dst, size_t inliers_capacity, int inliers_every,
                                     uint8_t* component_mask_dst, char* inliers_text_dst, size_t inliers_text_capacity);
/* one number as the device writes it (the same code built for the host): the characters of printf("%g", v) in out[0 .. return),
 * or -1 outside the formatter's domain.  out must hold 16 bytes.  Pure host function: no context, no GPU. */
int wass_format_g6(double v, char* out);
/* the selected inlier points of the frame whose wass_ctx_frame_result() was read last (n_inliers_out of them), copied on demand and
 * synchronously; valid until the next wass_mesh_finish_frame_async* call of the context */
int wass_ctx_frame_inliers(wass_ctx* ctx, double* dst, size_t capacity_points, uint64_t* n_out);
int wass_ctx_frame_result(wass_ctx* ctx, wass_frame_result* out);

/* RT_from_plane (:1044-1069); pure host math */
void wass_RT_from_plane(const double plane[4], double R[9], double T[3], double Rinv[9], double Tinv[3]);
/* save_as_xyz_compressed (:377-460): returns the exact bytes of mesh_cam.xyzC
 * in a malloc'ed host buffer (free with wass_free).  plane == NULL (RANSAC
 * failed): identity R, zero T are used (the reference reads uninitialised
 * memory there; documented divergence). */
int wass_mesh_encode_xyzc(wass_ctx* ctx, wass_mesh* m, const double plane[4], void** bytes, size_t* nbytes);
/* same, into a caller-owned buffer (>= 148 + 6*width*height bytes is always enough; pinned memory avoids a
 * staging copy) */
int wass_mesh_encode_xyzc_to(wass_ctx* ctx, wass_mesh* m, const double plane[4], void* dst, size_t capacity,
                             size_t* nbytes);
/* same, but returns as soon as the size is known: the 148-byte header is complete, the 6*n payload bytes are
 * being written by a DMA engine on the context's copy stream and are complete after wass_ctx_synchronize().
 * dst should be pinned host memory.  Lets a sequence driver overlap the PCIe transfer (and the file write of
 * wass_stereo.cpp:2123) with the next frame. */
int wass_mesh_encode_xyzc_async(wass_ctx* ctx, wass_mesh* m, const double plane[4], void* dst, size_t capacity,
                                size_t* nbytes);
void wass_free(void* p);

/* ---- row f3: the first step of the gridding stage from the device-resident mesh ----------------------------------
 * gridding/wassgridsurface/wassgridsurface.py:316-365 (_grid_task, "IDW"): align the cloud on the sequence's mean sea
 * plane (R, T = wass_RT_from_plane(mean plane), z negated: wass_utils.py:38-61), scale by the baseline in metres, bin the
 * points on the width x height grid over [xmin,xmax] x [ymin,ymax] (:322-326) and fill the gaps with
 * IDWInterpolator(KSIZE=5, exp=2.4, reps=1) (IDWInterpolator.py:23-58).  grid_out: height x width float32, NaN outside
 * the closed point mask; mask_out (may be NULL) that mask.  A cell takes the mean of its points (deterministic) where the
 * reference takes the median of ten random sub-samples (randomised): see grid.hip. */
typedef struct {
    double R[9], T[3];              /* gridsetup["Rpl"], ["Tpl"]                        */
    double baseline;                /* gridsetup["CAM_BASELINE"] (metres)               */
    double xmin, xmax, ymin, ymax;  /* grid extent in metres                            */
    int    width, height;           /* XX.shape[1], XX.shape[0]                         */
} wass_grid_setup;
int wass_mesh_grid_idw(wass_ctx* ctx, const wass_mesh* m, const wass_grid_setup* gs, float* grid_out, uint8_t* mask_out);
/* The same with the cell statistic chosen: WASS_GRID_CELL_MEAN (what wass_mesh_grid_idw computes) or WASS_GRID_CELL_MEDIAN, the
 * exact median of the points of a cell -- deterministic, independent of the point order, and robust against a few outliers
 * in a cell like the reference's nanmedian of random sub-samples (:330-345), whose expected value it is. */
enum { WASS_GRID_CELL_MEAN = 0, WASS_GRID_CELL_MEDIAN = 1 };
int wass_mesh_grid_idw_ex(wass_ctx* ctx, const wass_mesh* m, const wass_grid_setup* gs, int cell_statistic, float* grid_out,
                          uint8_t* mask_out);

/* Coll-1: NaN-aware mean of per-frame planes (np.nanmean of planes.txt,
 * gridding/wassgridsurface/wassgridsurface.py:672-678).  Reduces
 * [sum a, sum b, sum c, sum d, n_valid] into acc5 (caller all-reduces acc5
 * over ranks with RCCL/torch.distributed, then calls wass_planes_mean_finish). */
void wass_planes_mean_accumulate(const double* planes, int n, double acc5[5]);
void wass_planes_mean_finish(const double acc5[5], double mean_out[4], int* n_valid);

/* Coll-1 as a collective: one rank per GPU (one process per GPU, one context per process).  Rank 0 draws an id
 * (ncclGetUniqueId, 128 bytes) and hands it to the other ranks by whatever means the launcher has (pipe, file, env);
 * every rank then calls wass_coll_init and, once per sequence, wass_coll_allreduce_sum_f64 on the acc5 of
 * wass_planes_mean_accumulate -- an RCCL all-reduce (ncclSum, fp64) over xGMI on the context's stream, in place on
 * host values, followed by wass_planes_mean_finish.  librccl is loaded on first use; WASS_ERR_DEVICE if it is
 * missing.  Replaces: the shared output/planes.txt of cli/wasscli/wasscli.py:320,341-343 as the means of agreeing
 * on the sequence's mean plane.  The communicator is destroyed with the context. */
int wass_coll_unique_id(unsigned char id_out[128]);
int wass_coll_init(wass_ctx* ctx, int rank, int world, const unsigned char id[128]);
int wass_coll_allreduce_sum_f64(wass_ctx* ctx, double* values, int count /* <= 64 */);

/* ---- rectification: rectify() of wass_stereo.cpp:447-613 ------------------------
 * Rig-constant host math (no GPU work, usable without a context): */

/* cv::stereoRectify(K_left, 0, K_right, 0, size, R, T, R1, R2, P1, P2, Q, flags=0, alpha, size, &roi1, &roi2)
 * as called at wass_stereo.cpp:541 (Bouguet's algorithm, zero distortion).  3x3 / 3x4 row-major doubles,
 * roi = {x, y, width, height}.  Returns WASS_ERR_INVALID_ARG for a zero baseline. */
int wass_stereo_rectify(const double K_left[9], const double K_right[9], int width, int height, const double R[9],
                        const double T[3], double alpha, double R1[9], double R2[9], double P1[12], double P2[12],
                        int roi1[4], int roi2[4]);
/* cv::initUndistortRectifyMap(K, 0, R, P, size, CV_32FC1, map1, map2) (wass_stereo.cpp:600-601); maps are
 * [height][width] float32 source coordinates. */
int wass_init_rectify_map(const double K[9], const double R[9], const double P[12], int width, int height,
                          float* map_x, float* map_y);

/* Per-frame resampling on the GPU.  roi == NULL writes the full dw x dh image; otherwise only the
 * roi = {x, y, width, height} window of it is produced (the .clone() crops of wass_stereo.cpp:526-528,606-607
 * fused into the resampler) and dst is roi.width x roi.height, tightly packed. */

/* cv::remap(src, dst, map1, map2, cv::INTER_CUBIC) for CV_8UC1 / CV_32FC1 maps, BORDER_CONSTANT 0
 * (wass_stereo.cpp:603-604): 1/32-pixel coordinate quantisation, 15-bit fixed-point 4x4 weights. */
int wass_remap_cubic(wass_ctx* ctx, const uint8_t* src, int sw, int sh, size_t src_stride, const float* map_x,
                     const float* map_y, int dw, int dh, const int roi[4], uint8_t* dst);
int wass_remap_cubic_dev(wass_ctx* ctx, const uint8_t* d_src, int sw, int sh, size_t src_stride,
                         const float* d_map_x, const float* d_map_y, int dw, int dh, const int roi[4],
                         uint8_t* d_dst);
/* Row f2: cv::undistort(src, dst, K, dist) of wass_prepare (src/wass_prepare/wass_prepare.cpp:268): new camera
 * matrix = K, INTER_LINEAR, BORDER_CONSTANT 0, stripe-wise 1/32-pixel fixed-point maps.  dist = n_dist
 * coefficients in OpenCV order k1 k2 p1 p2 [k3 [k4 k5 k6 [s1 s2 s3 s4]]] (n_dist = 4, 5, 8 or 12; the tilt
 * model is not supported).  dst is w x h, tightly packed. */
int wass_undistort(wass_ctx* ctx, const uint8_t* src, int w, int h, size_t src_stride, const double K[9],
                   const double* dist, int n_dist, uint8_t* dst);
int wass_undistort_dev(wass_ctx* ctx, const uint8_t* d_src, int w, int h, size_t src_stride, const double K[9],
                       const double* dist, int n_dist, uint8_t* d_dst);
/* Row f2: cv::CLAHE::apply(src, dst) of wass_prepare (wass_prepare.cpp:257-262; createCLAHE(clip_limit, Size(tiles_x,
 * tiles_y)) at :446-449), CV_8UC1: clipped per-tile histograms, bilinear blend of the four neighbouring look-up tables.
 * dst is w x h, tightly packed. */
int wass_clahe(wass_ctx* ctx, const uint8_t* src, int w, int h, size_t src_stride, double clip_limit, int tiles_x, int tiles_y,
               uint8_t* dst);
int wass_clahe_dev(wass_ctx* ctx, const uint8_t* d_src, int w, int h, size_t src_stride, double clip_limit, int tiles_x, int tiles_y,
                   uint8_t* d_dst);
/* cv::warpPerspective(src, dst, H, Size(dw,dh)) with the default INTER_LINEAR / BORDER_CONSTANT 0
 * (wass_stereo.cpp:515-516); H maps source to destination pixels (it is inverted internally). */
int wass_warp_perspective(wass_ctx* ctx, const uint8_t* src, int sw, int sh, size_t src_stride, const double H[9],
                          int dw, int dh, const int roi[4], uint8_t* dst);
int wass_warp_perspective_dev(wass_ctx* ctx, const uint8_t* d_src, int sw, int sh, size_t src_stride,
                              const double H[9], int dw, int dh, const int roi[4], uint8_t* d_dst);

#ifdef __cplusplus
}
#endif
#endif
