/*
 * wass_oracle.h -- CPU restatement of the wass_stereo dense-stereo hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load it, and only as the checker / reported baseline.
 *
 * PARITY UNPINNED for the SGBM part: the arithmetic of cv::StereoSGBM lives
 * in OpenCV 4.5.5 (modules/calib3d/src/stereosgbm.cpp), a third-party
 * dependency that is neither vendored in /root/reference nor installed in
 * this image, and the reference has no golden disparity vectors.  This file
 * restates the published algorithm and is anchored on
 * the reference call sites src/wass_stereo/wass_stereo.cpp:775-782,837.
 * Everything else follows code that IS in the reference tree and cites it.
 *
 * Plain C99, no dependencies; compile with -ffp-contract=off so that the
 * fp64 geometry matches a non-FMA x86-64 build of the reference.
 */
#ifndef WASS_ORACLE_H
#define WASS_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- cv::StereoSGBM parameters as wass_stereo sets them
 *      (wass_stereo.cpp:772-782).  mode: 5 = MODE_SGBM, 8 = MODE_HH. ---- */
typedef struct {
    int min_disp;          /* MIN_DISPARITY            (wass_stereo.cpp:742) */
    int num_disp;          /* MAX_DISPARITY            (:743)                */
    int block_size;        /* WINSIZE                  (:744)                */
    int P1, P2;            /* DENSE_P{1,2}_MULT*w*w    (:772-773)            */
    int uniqueness_ratio;  /* DENSE_UNIQUENESS_RATIO   (:755,778)            */
    int disp12_max_diff;   /* DENSE_DISP12MAXDIFF      (:756,779)            */
    int prefilter_cap;     /* DENSE_PREFILTER_CAP      (:757,780)            */
    int speckle_window;    /* DENSE_SPECKLE_WINDOW_SIZE(:759,782) must be <=0*/
    int speckle_range;     /* DENSE_SPECKLE_RANGE      (:758,781)            */
    int mode;              /* 5 or 8                                         */
} orc_sgbm_params;

typedef struct {
    int max_C;             /* largest biased cost C (incl. +P2) seen, as int */
    int max_L;             /* largest path cost seen, as int                 */
    int overflow;          /* 1 if either exceeded 32767 (parity undefined)  */
} orc_sgbm_stats;

/* cv::StereoSGBM::compute(img1,img2,disp): img1/img2 are w x h u8 (pitch w),
 * disp16 is w x h int16 (fixed point, 4 fractional bits).
 * Optional dumps (may be NULL): C_out/S_out are [h][width1][D] int16 where
 * width1 = w - (min_disp+num_disp) (min_disp >= 0 assumed for the dump),
 * C_out is stored WITHOUT the +P2 bias. raw_out is disp before medianBlur. */
int orc_sgbm_compute(const uint8_t* img1, const uint8_t* img2, int w, int h,
                     const orc_sgbm_params* p, int16_t* disp16,
                     int16_t* C_out, int16_t* S_out, int16_t* raw_out,
                     orc_sgbm_stats* stats);

/* cv::medianBlur(ksize=3) on CV_16S, replicate border, out of place. */
void orc_median3_i16(const int16_t* src, int16_t* dst, int w, int h);

/* ---- wass_stereo.cpp:801-839: pad, compute(right,left), crop ---- */
int orc_dense_disparity16(const uint8_t* right, const uint8_t* left, int w, int h,
                          const orc_sgbm_params* p, int disparity_offset,
                          int16_t* disp16_crop, orc_sgbm_stats* stats);

/* ---- wass_stereo.cpp:714-733 ---- */
void orc_clean_and_convert(const int16_t* disp16, int w, int h, int mindisp,
                           int num_disp, int disp_offset, double scale, float* out);
/* ---- wass_stereo.cpp:617-662 / 665-711 (out of place) ---- */
void orc_dilate_zero(const float* src, float* out, int w, int h);
void orc_erode_zero(const float* src, float* out, int w, int h);
/* a7..a9 at DENSE_SCALE==1: convert, dilate x n, erode x m, then the
 * NN-erode mask of wass_stereo.cpp:903-928. */
void orc_disparity_postprocess(const int16_t* disp16, int w, int h, int mindisp,
                               int num_disp, int disp_offset, int dilate_steps,
                               int erode_steps, float* out);

/* cv::CLAHE::apply, CV_8UC1 (clahe_oracle.c; wass_prepare.cpp:257-262; PARITY UNPINNED) */
int orc_clahe(const uint8_t* src, int w, int h, size_t stride, double clip_limit, int tiles_x, int tiles_y, uint8_t* dst);

/* ---- optional parts of sgbm_dense_stereo, row a9 (a9_oracle.c; PARITY UNPINNED, OpenCV restated) ---- */
void orc_resize_dsize(int sw, int sh, double fx, double fy, int* dw, int* dh);          /* cv::resize(.., Size(), fx, fy) */
/* scale = source step per destination pixel (1/fx, or source size / destination size when dsize was given) */
void orc_resize_cubic_u8(const uint8_t* src, int sw, int sh, uint8_t* dst, int dw, int dh, double scale_x, double scale_y);
void orc_resize_cubic_f32(const float* src, int sw, int sh, float* dst, int dw, int dh, double scale_x, double scale_y);
void orc_resize_nn_f32(const float* src, int sw, int sh, float* dst, int dw, int dh, double scale_x, double scale_y);
size_t orc_biggest_component_by_gradient(float* disp, int w, int h, int threshold);     /* wass_stereo.cpp:947-986 */
void orc_filter_speckles(int16_t* img, int w, int h, int newVal, int maxSpeckleSize, int maxDiff);
void orc_disparity_postprocess_ex(const int16_t* disp16, int ws, int hs, int mindisp, int num_disp, int disp_offset,
                                  double dense_scale, int dilate_steps, int erode_steps, int cc_threshold, int ow, int oh,
                                  float* out);

/* ---- geometry for unrectify (wass_stereo.cpp:299-324) + triangulate ---- */
typedef struct {
    double K_left[9], K_right[9];   /* intrinsics (row major)              */
    double R[9], T[3];              /* env.R, env.T (|T| = 1)              */
    int    use_custom;              /* USE_CUSTOM_STEREORECTIFY            */
    double R1[9], R2[9];            /* rec_R1 / rec_R2                     */
    double P1[12], P2[12];          /* rec_P1 / rec_P2 (3x4)               */
    double HLi[9], HRi[9];          /* inverse homographies (custom path)  */
    double disparity_compensation;  /* env.disparity_compensation          */
    double dense_scale;             /* DENSE_SCALE                         */
} orc_geom;

typedef struct {
    double min_angle_deg;           /* TRIANG_MIN_ANGLE                    */
    double bbox[4];                 /* left, top, right, bottom (px, left) */
    double cam_distance;            /* env.cam_distance (=1)               */
} orc_tri_params;

/* wass_stereo.cpp:1039-1386.  disp is the full-frame float disparity (W x H,
 * the rectified right frame); left_rect/right_rect are only used for sizes;
 * left_img/right_img are the ORIGINAL undistorted images (for gray value and
 * sizes), masks are 0/1 u8 images of the same size as the originals.
 * mesh outputs are roi_r[2] x roi_r[3] grids: valid u8, p3d double[3], gray u8.
 * returns number of triangulated points. */
size_t orc_triangulate(const float* disp, int W, int H,
                       const int roi_l[4], const int roi_r[4],
                       const orc_geom* g,
                       const uint8_t* right_img, int img_w, int img_h,
                       const uint8_t* left_mask, const uint8_t* right_mask,
                       const orc_tri_params* tp,
                       uint8_t* valid, double* p3d, uint8_t* gray);

/* triangulate.hpp:26-72 */
void orc_triangulate_point(const double p[2], const double q[2],
                           const double R[9], const double T[3], double out[3]);

/* ---- PovMesh (SoA: valid[w*h], p3d[w*h*3]) ---- */
/* PovMesh.cpp:888-926; returns NaN if there are no gaps */
double orc_zgap_percentile(const uint8_t* valid, const double* p3d, int w, int h,
                           double percentile, size_t* n_gaps);
/* PovMesh.cpp:929-987 (+147-203); updates valid in place, returns size */
size_t orc_keep_biggest_component(uint8_t* valid, const double* p3d, int w, int h,
                                  double zgap);
/* sampler: PovMesh.cpp:680-691 with glibc rand(); caller srand()s first.
 * Fills uv[rounds][6] = {u1,v1,u2,v2,u3,v3}.  Argument evaluation order of
 * cv::Vec2i(rand()%W, rand()%H) follows GCC (right to left: v first). */
void orc_ransac_sample(int w, int h, int rounds, int32_t* uv);
/* PovMesh.cpp:665-777 given the samples; returns 1 on success (best >= N/10).
 * inliers_per_round (may be NULL) gets the count per round (-1 = skipped). */
int orc_ransac_plane(const uint8_t* valid, const double* p3d, int w, int h,
                     const int32_t* uv, int rounds, double thr,
                     double plane[4], size_t* best_inliers, int64_t* inliers_per_round);
/* PovMesh.cpp:780-815 */
size_t orc_crop_plane(uint8_t* valid, const double* p3d, int w, int h,
                      const double plane[4], double thr);
typedef struct {
    double xmin, xmax, ymin, ymax;  /* PLANE_REFINE_*                       */
    double max_distance;            /* PLANE_REFINEMENT_MAX_DISTANCE        */
    int weight_by_distance;         /* PLANE_WEIGHT_PROPORTIONAL_TO_DISTANCE*/
    int central_third_only;         /* PLANE_USE_CENTRAL_THIRD_ONLY         */
} orc_refine_params;
/* PovMesh.cpp:581-660; returns number of inliers; moments (may be NULL)
 * receives {wsum, c[3], A[9]} for reduction-level parity checks */
size_t orc_refine_plane(const uint8_t* valid, const double* p3d, int w, int h,
                        const orc_refine_params* rp, double plane[4], double* moments);
/* smallest-eigenvector of a symmetric 3x3 (stands in for cv::SVD vt row 2) */
void orc_smallest_eigvec3(const double A[9], double v[3]);
/* PovMesh.cpp:1044-1069 */
void orc_RT_from_plane(const double plane[4], double R[9], double T[3],
                       double Rinv[9], double Tinv[3]);
/* PovMesh.cpp:377-460; buf must hold 148 + 6*n_valid bytes; returns bytes */
size_t orc_encode_xyzc(const uint8_t* valid, const double* p3d, int w, int h,
                       const double plane[4], uint8_t* buf);

/* ---- rectification, row f1 (rectify_oracle.c; PARITY UNPINNED, OpenCV restated from knowledge) ---- */
/* initInterTab2D fixed-point weights: ksize 2 (bilinear) or 4 (bicubic); out = 1024*ksize*ksize int16 */
void orc_inter_tab(int ksize, int16_t* out);
/* cv::warpPerspective INTER_LINEAR / BORDER_CONSTANT 0 (wass_stereo.cpp:515-516) */
void orc_warp_perspective(const uint8_t* src, int sw, int sh, size_t src_stride, const double H[9], int dw, int dh,
                          uint8_t* dst);
/* cv::remap INTER_CUBIC, CV_32FC1 maps, BORDER_CONSTANT 0 (wass_stereo.cpp:603-604) */
void orc_remap_cubic(const uint8_t* src, int sw, int sh, size_t src_stride, const float* map_x, const float* map_y,
                     int dw, int dh, uint8_t* dst);
/* cv::initUndistortRectifyMap, zero distortion, CV_32FC1 (wass_stereo.cpp:600-601) */
int orc_init_rectify_map(const double K[9], const double R[9], const double P[12], int w, int h, float* map_x,
                         float* map_y);
/* cv::undistort(src, dst, K, dist) with INTER_LINEAR (wass_prepare.cpp:268), row f2; n = 4, 5, 8 or 12 coefficients */
int orc_undistort(const uint8_t* src, int w, int h, size_t src_stride, const double K[9], const double* dist, int n,
                  uint8_t* dst);
/* cv::stereoRectify flags=0 (wass_stereo.cpp:541) */
int orc_stereo_rectify(const double K1[9], const double K2[9], int W, int H, const double R[9], const double T[3],
                       double alpha, double R1[9], double R2[9], double P1[12], double P2[12], int roi1[4],
                       int roi2[4]);

#ifdef __cplusplus
}
#endif
#endif
