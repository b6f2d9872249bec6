/*
 * a9_oracle.c -- CPU restatement of the optional parts of sgbm_dense_stereo:
 *   DENSE_SCALE != 1             wass_stereo.cpp:788-796 (input resize), :853 (1/scale), :903-904 (output resize)
 *   biggest component by Sobel   :947-986
 *   filterSpeckles               inside cv::StereoSGBM::compute when DENSE_SPECKLE_WINDOW_SIZE > 0 (:758-759,781-782)
 *
 * TEST INFRASTRUCTURE ONLY (see wass_oracle.h).  PARITY UNPINNED: cv::resize, cv::Sobel, cv::connectedComponents and
 * cv::filterSpeckles are OpenCV 4.5.5 (modules/imgproc/src/resize.cpp, deriv.cpp / filter.simd.hpp, connectedcomponents.cpp,
 * modules/calib3d/src/stereosgbm.cpp), not in /root/reference and not installed; restated from the published
 * algorithms in their scalar form:
 *   resize:  INTER_CUBIC  -- source coordinate (dx + 0.5) * scale - 0.5, Keys kernel A = -0.75 evaluated in float, taps
 *            clamped to the image; CV_8U in 11-bit fixed point (coefficients rounded to short, horizontal pass in int,
 *            vertical pass (sum + 2^21) >> 22, saturated); CV_32F in float, taps accumulated left to right.
 *            INTER_NEAREST -- min(floor(dx * scale), size - 1).  scale = 1/fx when the call passes fx (dsize empty: the
 *            destination size is cvRound(size * fx) but the mapping keeps fx), source/destination size ratio when
 *            it passes dsize.
 *   Sobel:   3x3, BORDER_REFLECT_101, separable: [-1 0 1] then [1 2 1] (and transposed), float.
 *   components: 8-connected, labels numbered by the raster position of their first pixel; strictly largest area wins.
 *   speckles: 4-connected regions of pixels whose NEIGHBOURING values differ by at most maxDiff; regions of at most
 *            maxSpeckleSize pixels are set to newVal; pixels equal to newVal belong to no region.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "wass_oracle.h"

static int iclip(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
static int cv_round(double v) { return (int)lrint(v); }                 /* round half to even, like cvRound */

void orc_resize_dsize(int sw, int sh, double fx, double fy, int* dw, int* dh)
{
    *dw = cv_round(sw * fx);
    *dh = cv_round(sh * fy);
}

static void cubic_coeffs(float x, float* c)
{
    const float A = -0.75f;
    c[0] = ((A * (x + 1) - 5 * A) * (x + 1) + 8 * A) * (x + 1) - 4 * A;
    c[1] = ((A + 2) * x - (A + 3)) * x * x + 1;
    c[2] = ((A + 2) * (1 - x) - (A + 3)) * (1 - x) * (1 - x) + 1;
    c[3] = 1.f - c[0] - c[1] - c[2];
}

/* source position and weights of destination index d along one axis */
static void cubic_axis(int d, double scale, int* s0, float* c)
{
    float f = (float)((d + 0.5) * scale - 0.5);
    int s = (int)floorf(f);
    f -= (float)s;
    *s0 = s;
    cubic_coeffs(f, c);
}

void orc_resize_cubic_u8(const uint8_t* src, int sw, int sh, uint8_t* dst, int dw, int dh, double scale_x, double scale_y)
{
    int x, y, k, j;
    int* sx = (int*)malloc(sizeof(int) * (size_t)dw);
    short* ax = (short*)malloc(sizeof(short) * 4 * (size_t)dw);
    for (x = 0; x < dw; ++x) {
        float c[4];
        cubic_axis(x, scale_x, &sx[x], c);
        for (k = 0; k < 4; ++k) { int v = cv_round(c[k] * 2048.0f); ax[4 * x + k] = (short)iclip(v, -32768, 32767); }
    }
    for (y = 0; y < dh; ++y) {
        float c[4];
        int sy, by[4];
        cubic_axis(y, scale_y, &sy, c);
        for (k = 0; k < 4; ++k) by[k] = iclip(cv_round(c[k] * 2048.0f), -32768, 32767);
        for (x = 0; x < dw; ++x) {
            int acc = 0;
            for (k = 0; k < 4; ++k) {
                const uint8_t* row = src + (size_t)iclip(sy - 1 + k, 0, sh - 1) * sw;
                int hsum = 0;
                for (j = 0; j < 4; ++j) hsum += row[iclip(sx[x] - 1 + j, 0, sw - 1)] * ax[4 * x + j];
                acc += hsum * by[k];
            }
            dst[(size_t)y * dw + x] = (uint8_t)iclip((acc + (1 << 21)) >> 22, 0, 255);
        }
    }
    free(sx); free(ax);
}

void orc_resize_cubic_f32(const float* src, int sw, int sh, float* dst, int dw, int dh, double scale_x, double scale_y)
{
    int x, y, k;
    int* sx = (int*)malloc(sizeof(int) * (size_t)dw);
    float* ax = (float*)malloc(sizeof(float) * 4 * (size_t)dw);
    float* rows = (float*)malloc(sizeof(float) * 4 * (size_t)dw);
    for (x = 0; x < dw; ++x) cubic_axis(x, scale_x, &sx[x], &ax[4 * x]);
    for (y = 0; y < dh; ++y) {
        float c[4];
        int sy;
        cubic_axis(y, scale_y, &sy, c);
        for (k = 0; k < 4; ++k) {
            const float* row = src + (size_t)iclip(sy - 1 + k, 0, sh - 1) * sw;
            for (x = 0; x < dw; ++x) {
                const float* a = &ax[4 * x];
                rows[(size_t)k * dw + x] = row[iclip(sx[x] - 1, 0, sw - 1)] * a[0] + row[iclip(sx[x], 0, sw - 1)] * a[1] +
                                           row[iclip(sx[x] + 1, 0, sw - 1)] * a[2] + row[iclip(sx[x] + 2, 0, sw - 1)] * a[3];
            }
        }
        for (x = 0; x < dw; ++x)
            dst[(size_t)y * dw + x] = rows[x] * c[0] + rows[(size_t)dw + x] * c[1] + rows[2 * (size_t)dw + x] * c[2] + rows[3 * (size_t)dw + x] * c[3];
    }
    free(sx); free(ax); free(rows);
}

void orc_resize_nn_f32(const float* src, int sw, int sh, float* dst, int dw, int dh, double scale_x, double scale_y)
{
    int x, y;
    for (y = 0; y < dh; ++y) {
        const int sy = iclip((int)floor(y * scale_y), 0, sh - 1);
        for (x = 0; x < dw; ++x) dst[(size_t)y * dw + x] = src[(size_t)sy * sw + iclip((int)floor(x * scale_x), 0, sw - 1)];
    }
}

static int refl101(int i, int n) { if (n == 1) return 0; if (i < 0) i = -i; if (i >= n) i = 2 * n - 2 - i; return iclip(i, 0, n - 1); }

/* wass_stereo.cpp:947-986 in place; returns the area of the component kept (0: none) */
size_t orc_biggest_component_by_gradient(float* disp, int w, int h, int threshold)
{
    const size_t n = (size_t)w * h;
    size_t i, best_area = 0;
    int x, y, best = -1;
    float* gm = (float*)malloc(n * sizeof(float));
    int* label = (int*)malloc(n * sizeof(int));
    int* stack = (int*)malloc(n * sizeof(int));
    for (y = 0; y < h; ++y)
        for (x = 0; x < w; ++x) {
            float t[3], u[3], gx, gy;
            int k;
            for (k = 0; k < 3; ++k) {
                const float* r = disp + (size_t)refl101(y - 1 + k, h) * w;
                t[k] = r[refl101(x + 1, w)] - r[refl101(x - 1, w)];
                u[k] = r[refl101(x - 1, w)] + r[x] * 2 + r[refl101(x + 1, w)];
            }
            gx = t[0] + t[1] * 2 + t[2];
            gy = u[2] - u[0];
            gm[(size_t)y * w + x] = gx * gx + gy * gy;
        }
    for (i = 0; i < n; ++i) { if (gm[i] > (float)threshold) disp[i] = 0.0f; label[i] = 0; }
    {   /* 8-connected components of disp != 0, in raster order of their first pixel */
        int next = 0;
        for (i = 0; i < n; ++i) {
            size_t area = 0;
            int top = 0;
            if (disp[i] == 0.0f || label[i]) continue;
            ++next;
            label[i] = next; stack[top++] = (int)i;
            while (top) {
                const int p = stack[--top], px = p % w, py = p / w;
                int dx, dy;
                ++area;
                for (dy = -1; dy <= 1; ++dy)
                    for (dx = -1; dx <= 1; ++dx) {
                        const int qx = px + dx, qy = py + dy;
                        size_t q;
                        if ((!dx && !dy) || qx < 0 || qy < 0 || qx >= w || qy >= h) continue;
                        q = (size_t)qy * w + qx;
                        if (disp[q] != 0.0f && !label[q]) { label[q] = next; stack[top++] = (int)q; }
                    }
            }
            if (area > best_area) { best_area = area; best = next; }
        }
    }
    for (i = 0; i < n; ++i) if (label[i] != best) disp[i] = 0.0f;
    free(gm); free(label); free(stack);
    return best_area;
}

/* cv::filterSpeckles(img, newVal, maxSpeckleSize, maxDiff) on CV_16S, in place */
void orc_filter_speckles(int16_t* img, int w, int h, int newVal, int maxSpeckleSize, int maxDiff)
{
    const size_t n = (size_t)w * h;
    size_t i;
    int* label = (int*)calloc(n, sizeof(int));
    int* stack = (int*)malloc(n * sizeof(int));
    int next = 0;
    for (i = 0; i < n; ++i) {
        int top = 0, count = 0;
        if (img[i] == newVal || label[i]) continue;
        ++next;
        label[i] = next; stack[top++] = (int)i;
        while (top) {
            const int p = stack[--top], px = p % w, py = p / w, dp = img[p];
            static const int DX[4] = { 1, -1, 0, 0 }, DY[4] = { 0, 0, 1, -1 };
            int k;
            ++count;
            for (k = 0; k < 4; ++k) {
                const int qx = px + DX[k], qy = py + DY[k];
                size_t q;
                if (qx < 0 || qy < 0 || qx >= w || qy >= h) continue;
                q = (size_t)qy * w + qx;
                if (!label[q] && img[q] != newVal && abs(dp - img[q]) <= maxDiff) { label[q] = next; stack[top++] = (int)q; }
            }
        }
        if (count <= maxSpeckleSize) {           /* second sweep over the region: it is small */
            size_t q;
            for (q = 0; q < n; ++q) if (label[q] == next) img[q] = (int16_t)newVal;
        }
    }
    free(label); free(stack);
}

/* sgbm_dense_stereo after compute(): wass_stereo.cpp:853-986 with every option.  disp16 is ws x hs (the resized input
 * size), out is ow x oh (roi_comb_right.size()).  dense_scale == 1 and ow x oh == ws x hs reproduce orc_disparity_postprocess. */
void orc_disparity_postprocess_ex(const int16_t* disp16, int ws, int hs, int mindisp, int num_disp, int disp_offset,
                                  double dense_scale, int dilate_steps, int erode_steps, int cc_threshold, int ow, int oh, float* out)
{
    const size_t n = (size_t)ws * hs, no = (size_t)ow * oh;
    size_t i;
    int s;
    float* a = (float*)malloc(n * sizeof(float));
    float* b = (float*)malloc(n * sizeof(float));
    float* nn = (float*)malloc(no * sizeof(float));
    float* ne = (float*)malloc(no * sizeof(float));
    orc_clean_and_convert(disp16, ws, hs, mindisp, num_disp, disp_offset, 1.0 / dense_scale, a);
    for (s = 1; s <= dilate_steps; ++s) { orc_dilate_zero(a, b, ws, hs); memcpy(a, b, n * sizeof(float)); }
    for (s = 1; s <= erode_steps; ++s) { orc_erode_zero(a, b, ws, hs); memcpy(a, b, n * sizeof(float)); }
    if (ow == ws && oh == hs) {                  /* cv::resize to the same size is a copy */
        memcpy(nn, a, no * sizeof(float)); memcpy(out, a, no * sizeof(float));
    } else {
        orc_resize_nn_f32(a, ws, hs, nn, ow, oh, (double)ws / ow, (double)hs / oh);
        orc_resize_cubic_f32(a, ws, hs, out, ow, oh, (double)ws / ow, (double)hs / oh);
    }
    orc_erode_zero(nn, ne, ow, oh);
    for (i = 0; i < no; ++i) if (ne[i] == 0) out[i] = 0.0f;
    if (cc_threshold > 0) orc_biggest_component_by_gradient(out, ow, oh, cc_threshold);
    free(a); free(b); free(nn); free(ne);
}
