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
    WASS_ERR_UNSUPPORTED = -2,   /* e.g. speckle filter on, DENSE_SCALE != 1   */
    WASS_ERR_NO_MEMORY = -3,
    WASS_ERR_DEVICE = -4,        /* HIP runtime error, no GPU                  */
    WASS_ERR_COST_OVERFLOW = -5, /* int16 cost precondition violated (A.7)     */
    WASS_ERR_TOO_FEW_POINTS = -6
} wass_status;

typedef struct wass_ctx wass_ctx;

int wass_ctx_create(int device_id, wass_ctx** out);
void wass_ctx_destroy(wass_ctx* ctx);
const char* wass_last_error(const wass_ctx* ctx);
/* raw hipStream_t of the context (for callers that enqueue their own work) */
void* wass_ctx_stream(wass_ctx* ctx);
int wass_ctx_synchronize(wass_ctx* ctx);
/* test mode: keep intermediates (the finished S volume) that production never writes to HBM */
int wass_ctx_set_debug(wass_ctx* ctx, int on);
const char* wass_version(void);

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
    int speckle_win;      /* DENSE_SPECKLE_WINDOW_SIZE (must be <= 0)        */
    int speckle_range;    /* DENSE_SPECKLE_RANGE                             */
    int ndirs;            /* 5 or 8                                          */
    int disp_offset;      /* DISPARITY_OFFSET (:747,801-812)                 */
    double dense_scale;   /* DENSE_SCALE (:745); only 1.0 is supported       */
} wass_sgm_params;

/* Replaces wass_stereo.cpp:820-839: zero-pad both rectified crops, run
 * dense_stereo->compute(right_image, left_image, disparity), crop columns
 * [num_disp, num_disp + w).  right/left: w x h u8, pitch in bytes.
 * disp16_out: w x h int16 (4 fractional bits), in the right image's frame.
 * Returns WASS_ERR_COST_OVERFLOW (result still written) if a block cost
 * exceeded the int16 range the reference's scalar and SIMD builds agree on. */
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
} wass_sgm_timings;
int wass_sgm_last_timings(wass_ctx* ctx, wass_sgm_timings* out);

/* Test hooks: copy intermediates of the last wass_sgm_disparity call to host.
 * C/S are [h][width1][num_disp] int16 with width1 = w + max(disp_offset,0) -
 * min_disp (C without the +P2 bias); raw is the padded-width disparity before
 * the 3x3 median ([h][w + num_disp + max(disp_offset,0)]).  Any may be NULL.
 * S_out requires wass_ctx_set_debug(ctx, 1) before the disparity call. */
int wass_sgm_debug_fetch(wass_ctx* ctx, int16_t* C_out, int16_t* S_out, int16_t* raw_out);

#ifdef __cplusplus
}
#endif
#endif
