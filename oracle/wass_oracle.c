/*
 * wass_oracle.c -- CPU restatement of the parts of the wass_stereo hot path
 * that live in the reference tree itself (everything around cv::StereoSGBM).
 *
 * TEST INFRASTRUCTURE ONLY (see wass_oracle.h).  Each function cites the
 * reference file:line it follows; paths are relative to /root/reference/src.
 */
#include "wass_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------ */
/* wass_stereo/wass_stereo.cpp:801-839: zero-pad both images to width
 * cols + D + max(offset,0), call compute(right_image, left_image), crop
 * columns [D, D+cols).  DENSE_SCALE == 1 (identity).     */
int orc_dense_disparity16(const uint8_t* right, const uint8_t* left, int w, int h,
                          const orc_sgbm_params* p, int disparity_offset,
                          int16_t* disp16_crop, orc_sgbm_stats* stats)
{
    const int D = p->num_disp;
    const int off = disparity_offset > 0 ? disparity_offset : 0;
    const int comp = disparity_offset > 0 ? 0 : -disparity_offset;
    const int W = w + D + off;
    uint8_t* L = (uint8_t*)calloc((size_t)W * h, 1);
    uint8_t* R = (uint8_t*)calloc((size_t)W * h, 1);
    int16_t* disp = (int16_t*)malloc((size_t)W * h * sizeof(int16_t));
    int y, rc;
    if (!L || !R || !disp) return -3;
    if (D + off - comp < 0) { free(L); free(R); free(disp); return -5; } /* colRange would throw */
    for (y = 0; y < h; y++) {
        memcpy(L + (size_t)y * W + (D + off - comp), left + (size_t)y * w, (size_t)w);
        memcpy(R + (size_t)y * W + D, right + (size_t)y * w, (size_t)w);
    }
    rc = orc_sgbm_compute(R, L, W, h, p, disp, NULL, NULL, NULL, stats);
    if (rc == 0)
        for (y = 0; y < h; y++)
            memcpy(disp16_crop + (size_t)y * w, disp + (size_t)y * W + D, (size_t)w * sizeof(int16_t));
    free(L); free(R); free(disp);
    return rc;
}

/* wass_stereo/wass_stereo.cpp:714-733 */
void orc_clean_and_convert(const int16_t* disp16, int w, int h, int mindisp,
                           int num_disp, int disp_offset, double scale, float* out)
{
    size_t i, n = (size_t)w * h;
    for (i = 0; i < n; i++) {
        float dval = ((float)disp16[i]) / 16.0f;
        out[i] = 0.0f;
        if (dval <= mindisp || dval > num_disp) continue;
        dval += disp_offset;
        out[i] = (float)(dval * scale);
    }
}

/* wass_stereo/wass_stereo.cpp:617-662.  The output pointer starts at column
 * 0 while the stencil is centred on column 1: pixel (i,k), k in [0,w-3], is
 * filled from the 8-neighbourhood of (i,k+1) (which never includes (i,k+1)). */
void orc_dilate_zero(const float* src, float* out, int w, int h)
{
    int i, j;
    memcpy(out, src, (size_t)w * h * sizeof(float));
    for (i = 1; i < h - 1; ++i) {
        const float* t = src + (size_t)(i - 1) * w + 1;
        const float* b = src + (size_t)(i + 1) * w + 1;
        const float* c = src + (size_t)i * w + 1;
        float* o = out + (size_t)i * w;
        for (j = 1; j < w - 1; ++j, ++t, ++b, ++c, ++o) {
            if (*o == 0) {
                float avg = 0; int n = 0;
                if (t[-1] > 0) { avg += t[-1]; ++n; }
                if (t[1] > 0) { avg += t[1]; ++n; }
                if (t[0] > 0) { avg += t[0]; ++n; }
                if (b[-1] > 0) { avg += b[-1]; ++n; }
                if (b[1] > 0) { avg += b[1]; ++n; }
                if (b[0] > 0) { avg += b[0]; ++n; }
                if (c[-1] > 0) { avg += c[-1]; ++n; }
                if (c[1] > 0) { avg += c[1]; ++n; }
                if (n > 1) *o = avg / (float)n;
            }
        }
    }
}

/* wass_stereo/wass_stereo.cpp:665-711 */
void orc_erode_zero(const float* src, float* out, int w, int h)
{
    int i, j;
    memcpy(out, src, (size_t)w * h * sizeof(float));
    for (i = 1; i < h - 1; ++i) {
        const float* t = src + (size_t)(i - 1) * w;
        const float* b = src + (size_t)(i + 1) * w;
        const float* c = src + (size_t)i * w;
        float* o = out + (size_t)i * w;
        o[0] = 0;
        for (j = 1; j < w - 1; ++j)
            if (t[j] == 0 || t[j - 1] == 0 || t[j + 1] == 0 || b[j] == 0 || b[j - 1] == 0 ||
                b[j + 1] == 0 || c[j - 1] == 0 || c[j + 1] == 0)
                o[j] = 0;
        /* the reference zeroes "the last column" through a pointer that sits on
         * column w-1 only when w >= 2; for w == 1 it re-zeroes column 0+1 (OOB) */
        if (w >= 2) o[w - 1] = 0;
    }
    for (j = 0; j < w; ++j) { out[j] = 0; out[(size_t)(h - 1) * w + j] = 0; }
}

/* wass_stereo/wass_stereo.cpp:853-928 at DENSE_SCALE == 1: both cv::resize
 * calls are same-size copies, the NN copy is eroded once more and the cubic
 * copy is zeroed where the eroded NN copy is zero. */
void orc_disparity_postprocess(const int16_t* disp16, int w, int h, int mindisp,
                               int num_disp, int disp_offset, int dilate_steps,
                               int erode_steps, float* out)
{
    size_t n = (size_t)w * h, i;
    float* a = (float*)malloc(n * sizeof(float));
    float* b = (float*)malloc(n * sizeof(float));
    int s;
    orc_clean_and_convert(disp16, w, h, mindisp, num_disp, disp_offset, 1.0 / 1.0, a);
    for (s = 1; s <= dilate_steps; ++s) { orc_dilate_zero(a, b, w, h); memcpy(a, b, n * sizeof(float)); }
    for (s = 1; s <= erode_steps; ++s) { orc_erode_zero(a, b, w, h); memcpy(a, b, n * sizeof(float)); }
    orc_erode_zero(a, b, w, h);                 /* :908-910 on the NN copy */
    for (i = 0; i < n; i++) out[i] = (b[i] == 0) ? 0.0f : a[i];   /* :914-928 */
    free(a); free(b);
}

/* ------------------------------------------------------------------------ */
static void mat3_mulv(const double M[9], const double v[3], double o[3])
{
    o[0] = M[0] * v[0] + M[1] * v[1] + M[2] * v[2];
    o[1] = M[3] * v[0] + M[4] * v[1] + M[5] * v[2];
    o[2] = M[6] * v[0] + M[7] * v[1] + M[8] * v[2];
}
static void mat3_tmulv(const double M[9], const double v[3], double o[3])
{
    o[0] = M[0] * v[0] + M[3] * v[1] + M[6] * v[2];
    o[1] = M[1] * v[0] + M[4] * v[1] + M[7] * v[2];
    o[2] = M[2] * v[0] + M[5] * v[1] + M[8] * v[2];
}

/* wass_stereo/wass_stereo.cpp:299-324 */
static void unrectify(const orc_geom* g, double u, double v, int use_left, double out[2])
{
    if (g->use_custom) {
        const double* Hi = use_left ? g->HLi : g->HRi;
        double in[3] = { u, v, 1.0 }, r[3];
        mat3_mulv(Hi, in, r);
        out[0] = r[0] / r[2]; out[1] = r[1] / r[2];
    } else {
        const double* K = use_left ? g->K_left : g->K_right;
        const double* R = use_left ? g->R1 : g->R2;
        const double* P = use_left ? g->P1 : g->P2;   /* 3x4 */
        double xyw[3] = { (u - P[2]) / P[0], (v - P[6]) / P[5], 1.0 }, r[3];
        mat3_tmulv(R, xyw, r);
        r[0] /= r[2]; r[1] /= r[2];
        out[0] = r[0] * K[0] + K[2];
        out[1] = r[1] * K[4] + K[5];
    }
}

/* wass_lib/triangulate.hpp:26-72; cv::solve(3x3, DECOMP_LU) restated as the
 * closed-form determinant solve OpenCV uses for 3x3 systems. */
void orc_triangulate_point(const double p[2], const double q[2],
                           const double R[9], const double T[3], double out[3])
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

    {
        double det = A[0] * (A[4] * A[8] - A[5] * A[7]) - A[1] * (A[3] * A[8] - A[5] * A[6]) +
                     A[2] * (A[3] * A[7] - A[4] * A[6]);
        if (det != 0) {
            double t0, t1, t2;
            det = 1. / det;
            t0 = det * (b[0] * (A[4] * A[8] - A[5] * A[7]) - A[1] * (b[1] * A[8] - A[5] * b[2]) +
                        A[2] * (b[1] * A[7] - A[4] * b[2]));
            t1 = det * (A[0] * (b[1] * A[8] - A[5] * b[2]) - b[0] * (A[3] * A[8] - A[5] * A[6]) +
                        A[2] * (A[3] * b[2] - b[1] * A[6]));
            t2 = det * (A[0] * (A[4] * b[2] - b[1] * A[7]) - A[1] * (A[3] * b[2] - b[1] * A[6]) +
                        b[0] * (A[3] * A[7] - A[4] * A[6]));
            out[0] = t0; out[1] = t1; out[2] = t2;
        } else {
            out[0] = out[1] = out[2] = 0.0;   /* cv::solve leaves x untouched and returns false */
        }
    }
}

/* wass_stereo/wass_stereo.cpp:1039-1386 */
size_t orc_triangulate(const float* disp, int W, int H,
                       const int roi_l[4], const int roi_r[4],
                       const orc_geom* g,
                       const uint8_t* right_img, int img_w, int img_h,
                       const uint8_t* left_mask, const uint8_t* right_mask,
                       const orc_tri_params* tp,
                       uint8_t* valid, double* p3d, uint8_t* gray)
{
    const int min_disp = 1;                                   /* :1100 */
    const int mw = roi_r[2], mh = roi_r[3];
    size_t n_pts = 0;
    int yr, xr;
    memset(valid, 0, (size_t)mw * mh);
    memset(p3d, 0, (size_t)mw * mh * 3 * sizeof(double));
    memset(gray, 0, (size_t)mw * mh);

    for (yr = roi_r[1]; yr < roi_r[1] + roi_r[3]; yr++) {
        for (xr = roi_r[0]; xr < roi_r[0] + roi_r[2]; xr++) {
            const float dv = disp[(size_t)yr * W + xr];
            float xl, yl;
            double pi[2], qi[2], p[2], q[2], P[3], dist;
            int skip = 0;
            if (!(dv > min_disp)) continue;                    /* :1177 */
            /* :1180 -- the sum is evaluated in double, then narrowed */
            xl = (float)(xr - roi_r[0] + roi_l[0] - dv + g->disparity_compensation / g->dense_scale);
            yl = (float)yr;
            if (xl < 0 || xl >= W) continue;                   /* :1183 (left_rectified.cols == W) */

            unrectify(g, (double)xl, (double)yl, 1, pi);       /* :1219 */
            unrectify(g, (double)xr, (double)yr, 0, qi);       /* :1220 */

            if (pi[0] < 1 || pi[0] >= img_w - 1 || pi[1] < 1 || pi[1] >= img_h - 1 ||
                qi[0] < 1 || qi[0] >= img_w - 1 || qi[1] < 1 || qi[1] >= img_h - 1)
                skip = 1;                                      /* :1223 */

            p[0] = (pi[0] - g->K_left[2]) / g->K_left[0];      /* :1232 */
            p[1] = (pi[1] - g->K_left[5]) / g->K_left[4];
            q[0] = (qi[0] - g->K_right[2]) / g->K_right[0];    /* :1233 */
            q[1] = (qi[1] - g->K_right[5]) / g->K_right[4];

            if (pi[0] <= tp->bbox[0] || pi[1] <= tp->bbox[1] ||
                pi[0] >= tp->bbox[2] || pi[1] >= tp->bbox[3])
                skip = 1;                                      /* :1236 */

            /* :1244,1250 -- the reference indexes the masks even when the point is
             * outside the image (UB); short-circuit as SURVEY.md a11 prescribes */
            if (!skip) {
                if (left_mask && left_mask[(size_t)(int)pi[1] * img_w + (int)pi[0]] == 0) skip = 1;
                if (right_mask && right_mask[(size_t)(int)qi[1] * img_w + (int)qi[0]] == 0) skip = 1;
            }

            if (tp->min_angle_deg > 0) {                       /* :1258-1269 */
                double d1[3] = { p[0], p[1], 1.0 }, qq[3] = { q[0], q[1], 1.0 }, d2[3], n1, n2, ang;
                mat3_mulv(g->R, qq, d2);
                d2[0] += g->T[0]; d2[1] += g->T[1]; d2[2] += g->T[2];
                n1 = sqrt(d1[0] * d1[0] + d1[1] * d1[1] + d1[2] * d1[2]);
                n2 = sqrt(d2[0] * d2[0] + d2[1] * d2[1] + d2[2] * d2[2]);
                d1[0] /= n1; d1[1] /= n1; d1[2] /= n1;
                d2[0] /= n2; d2[1] /= n2; d2[2] /= n2;
                ang = fabs(acos(d1[0] * d2[0] + d1[1] * d2[1] + d1[2] * d2[2]) * 57.29577951);
                if (ang < tp->min_angle_deg) skip = 1;
            }
            if (skip) continue;                                /* :1272 */

            orc_triangulate_point(p, q, g->R, g->T, P);        /* :1286 */

            dist = sqrt(P[0] * P[0] + P[1] * P[1] + P[2] * P[2]);
            if (dist < tp->cam_distance / 10.0 || P[2] < 1.0) continue;       /* :1329 */
            if (dist > tp->cam_distance * 200.0 || P[2] > 1E30) continue;     /* :1335 */

            {
                const int u = xr - roi_r[0], v = yr - roi_r[1];                /* :1345 */
                const size_t idx = (size_t)v * mw + u;
                valid[idx] = 1;
                p3d[idx * 3 + 0] = P[0]; p3d[idx * 3 + 1] = P[1]; p3d[idx * 3 + 2] = P[2];
                gray[idx] = right_img[(size_t)(int)qi[1] * img_w + (int)qi[0]]; /* :1342 */
            }
            n_pts++;
        }
    }
    return n_pts;
}

/* ------------------------------------------------------------------------ */
static int cmp_double(const void* a, const void* b)
{
    double x = *(const double*)a, y = *(const double*)b;
    return (x > y) - (x < y);
}

/* wass_stereo/PovMesh.cpp:888-926.  Index clamped to n-1 (the reference reads
 * out of bounds at percentile == 100). */
double orc_zgap_percentile(const uint8_t* valid, const double* p3d, int w, int h,
                           double percentile, size_t* n_gaps)
{
    double* g = (double*)malloc((size_t)w * h * 3 * sizeof(double) + 8);
    size_t n = 0, k;
    double r;
    int i, j;
    for (i = 1; i < h; i++)
        for (j = 1; j < w - 1; j++) {
            const size_t c = (size_t)i * w + j;
            if (!valid[c]) continue;
            {
                const size_t A = c - w - 1, B = c - w, C = c - w + 1;
                if (valid[A]) g[n++] = fabs(p3d[c * 3 + 2] - p3d[A * 3 + 2]);
                if (valid[B]) g[n++] = fabs(p3d[c * 3 + 2] - p3d[B * 3 + 2]);
                if (valid[C]) g[n++] = fabs(p3d[c * 3 + 2] - p3d[C * 3 + 2]);
            }
        }
    if (n_gaps) *n_gaps = n;
    if (n == 0) { free(g); return NAN; }
    qsort(g, n, sizeof(double), cmp_double);
    k = (size_t)floor(percentile / 100.0 * (double)n);
    if (k >= n) k = n - 1;
    r = g[k];
    free(g);
    return r;
}

/* wass_stereo/PovMesh.cpp:929-987 with get_non_visited_neighbours (:147-188) and
 * get_non_visited (:190-203).  Seeds are found in column-major order; scanning
 * resumes where the previous seed was found (equivalent: visited only grows). */
size_t orc_keep_biggest_component(uint8_t* valid, const double* p3d, int w, int h, double zgap)
{
    const size_t N = (size_t)w * h;
    int32_t* comp = (int32_t*)malloc(N * sizeof(int32_t));
    uint8_t* visited = (uint8_t*)calloc(N, 1);
    size_t* stack = (size_t*)malloc((N * 4 + 16) * sizeof(size_t));
    size_t remaining = 0, biggest = 0, i, sp;
    int cur = 0, best = 0, su = 0, sv = 0;
    for (i = 0; i < N; i++) { comp[i] = -1; remaining += valid[i] ? 1 : 0; }

    for (;;) {
        size_t size = 0;
        int found = 0;
        for (; su < w && !found; ) {
            for (; sv < h; ++sv) {
                size_t c = (size_t)sv * w + su;
                if (valid[c] && !visited[c]) { found = 1; break; }
            }
            if (!found) { ++su; sv = 0; }
        }
        if (!found) break;
        sp = 0;
        stack[sp++] = (size_t)sv * w + su;
        while (sp) {
            size_t c = stack[--sp];
            int u = (int)(c % w), v = (int)(c / w);
            double z;
            if (visited[c]) continue;
            visited[c] = 1; comp[c] = cur; size++;
            z = p3d[c * 3 + 2];
            if (u > 0 && valid[c - 1] && !visited[c - 1] && fabs(z - p3d[(c - 1) * 3 + 2]) < zgap) stack[sp++] = c - 1;
            if (u < w - 1 && valid[c + 1] && !visited[c + 1] && fabs(z - p3d[(c + 1) * 3 + 2]) < zgap) stack[sp++] = c + 1;
            if (v > 0 && valid[c - w] && !visited[c - w] && fabs(z - p3d[(c - w) * 3 + 2]) < zgap) stack[sp++] = c - w;
            if (v < h - 1 && valid[c + w] && !visited[c + w] && fabs(z - p3d[(c + w) * 3 + 2]) < zgap) stack[sp++] = c + w;
        }
        if (size > biggest) { biggest = size; best = cur; }
        remaining -= size;
        if (remaining < biggest) break;
        cur++;
    }
    for (i = 0; i < N; i++) valid[i] = (comp[i] == best) ? 1 : 0;  /* extract_component (:96-100) */
    free(comp); free(visited); free(stack);
    return biggest;
}

/* wass_stereo/PovMesh.cpp:680-691 */
void orc_ransac_sample(int w, int h, int rounds, int32_t* uv)
{
    const double mindist = h * 0.01;
    int r = 0;
    while (r < rounds) {
        int c[6], k;
        double d12, d23, d13;
        for (k = 0; k < 3; k++) {      /* GCC evaluates (rand()%W, rand()%H) right to left */
            int v = rand() % h;
            int u = rand() % w;
            c[2 * k] = u; c[2 * k + 1] = v;
        }
        d12 = sqrt((double)(c[0] - c[2]) * (c[0] - c[2]) + (double)(c[1] - c[3]) * (c[1] - c[3]));
        d23 = sqrt((double)(c[2] - c[4]) * (c[2] - c[4]) + (double)(c[3] - c[5]) * (c[3] - c[5]));
        d13 = sqrt((double)(c[0] - c[4]) * (c[0] - c[4]) + (double)(c[1] - c[5]) * (c[1] - c[5]));
        if (d12 < mindist || d23 < mindist || d13 < mindist) continue;   /* round--; continue */
        for (k = 0; k < 6; k++) uv[(size_t)r * 6 + k] = c[k];
        r++;
    }
}

/* wass_stereo/PovMesh.cpp:665-777 */
int orc_ransac_plane(const uint8_t* valid, const double* p3d, int w, int h,
                     const int32_t* uv, int rounds, double thr,
                     double plane[4], size_t* best_inliers, int64_t* inliers_per_round)
{
    const size_t N = (size_t)w * h;
    size_t best = 0, i;
    double bn[3] = { 0, 0, 0 }, bd = 0;
    int r;
    for (r = 0; r < rounds; r++) {
        const int32_t* c = uv + (size_t)r * 6;
        const size_t i1 = (size_t)c[1] * w + c[0], i2 = (size_t)c[3] * w + c[2], i3 = (size_t)c[5] * w + c[4];
        double a[3], b[3], n[3], nn, d;
        size_t cnt = 0;
        if (inliers_per_round) inliers_per_round[r] = -1;
        if (!valid[i1] || !valid[i2] || !valid[i3]) continue;
        a[0] = p3d[i2 * 3] - p3d[i1 * 3]; a[1] = p3d[i2 * 3 + 1] - p3d[i1 * 3 + 1]; a[2] = p3d[i2 * 3 + 2] - p3d[i1 * 3 + 2];
        b[0] = p3d[i3 * 3] - p3d[i1 * 3]; b[1] = p3d[i3 * 3 + 1] - p3d[i1 * 3 + 1]; b[2] = p3d[i3 * 3 + 2] - p3d[i1 * 3 + 2];
        n[0] = a[1] * b[2] - a[2] * b[1];
        n[1] = a[2] * b[0] - a[0] * b[2];
        n[2] = a[0] * b[1] - a[1] * b[0];
        nn = sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
        n[0] = n[0] / nn; n[1] = n[1] / nn; n[2] = n[2] / nn;
        if (n[2] < 0) { n[0] = n[0] * -1.0; n[1] = n[1] * -1.0; n[2] = n[2] * -1.0; }
        d = -(n[0] * p3d[i1 * 3] + n[1] * p3d[i1 * 3 + 1] + n[2] * p3d[i1 * 3 + 2]);
        for (i = 0; i < N; i++)
            if (valid[i]) {
                double dist = fabs((n[0] * p3d[i * 3] + n[1] * p3d[i * 3 + 1] + n[2] * p3d[i * 3 + 2]) + d);
                if (dist < thr) ++cnt;
            }
        if (inliers_per_round) inliers_per_round[r] = (int64_t)cnt;
        if (cnt > best) { best = cnt; bn[0] = n[0]; bn[1] = n[1]; bn[2] = n[2]; bd = d; }
    }
    plane[0] = bn[0]; plane[1] = bn[1]; plane[2] = bn[2]; plane[3] = bd;
    if (best_inliers) *best_inliers = best;
    return best < N / 10 ? 0 : 1;
}

/* wass_stereo/PovMesh.cpp:780-815 */
size_t orc_crop_plane(uint8_t* valid, const double* p3d, int w, int h,
                      const double plane[4], double thr)
{
    const size_t N = (size_t)w * h;
    size_t i, k = 0;
    for (i = 0; i < N; i++) {
        if (valid[i]) {
            double d = fabs((plane[0] * p3d[i * 3] + plane[1] * p3d[i * 3 + 1] + plane[2] * p3d[i * 3 + 2]) + plane[3]);
            if (d < thr) { ++k; } else valid[i] = 0;
        }
    }
    return k;
}

/* Jacobi eigen-decomposition of a symmetric 3x3; returns the unit eigenvector
 * of the smallest eigenvalue (= row 2 of vt in cv::SVD for a PSD matrix). */
void orc_smallest_eigvec3(const double Ain[9], double vout[3])
{
    double A[3][3], V[3][3] = { { 1, 0, 0 }, { 0, 1, 0 }, { 0, 0, 1 } };
    int it, i, j, k, m = 0;
    for (i = 0; i < 3; i++) for (j = 0; j < 3; j++) A[i][j] = Ain[i * 3 + j];
    for (it = 0; it < 64; it++) {
        double off = fabs(A[0][1]) + fabs(A[0][2]) + fabs(A[1][2]);
        double diag = fabs(A[0][0]) + fabs(A[1][1]) + fabs(A[2][2]);
        if (off <= 1e-300 || off <= 1e-22 * diag) break;
        for (i = 0; i < 2; i++)
            for (j = i + 1; j < 3; j++) {
                double theta, t, c, s;
                if (A[i][j] == 0.0) continue;
                theta = (A[j][j] - A[i][i]) / (2.0 * A[i][j]);
                t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                c = 1.0 / sqrt(t * t + 1.0); s = t * c;
                for (k = 0; k < 3; k++) {
                    double aik = A[i][k], ajk = A[j][k];
                    A[i][k] = c * aik - s * ajk; A[j][k] = s * aik + c * ajk;
                }
                for (k = 0; k < 3; k++) {
                    double aki = A[k][i], akj = A[k][j];
                    A[k][i] = c * aki - s * akj; A[k][j] = s * aki + c * akj;
                }
                for (k = 0; k < 3; k++) {
                    double vki = V[k][i], vkj = V[k][j];
                    V[k][i] = c * vki - s * vkj; V[k][j] = s * vki + c * vkj;
                }
            }
    }
    for (i = 1; i < 3; i++) if (A[i][i] < A[m][m]) m = i;
    {
        double n = sqrt(V[0][m] * V[0][m] + V[1][m] * V[1][m] + V[2][m] * V[2][m]);
        vout[0] = V[0][m] / n; vout[1] = V[1][m] / n; vout[2] = V[2][m] / n;
    }
}

/* wass_stereo/PovMesh.cpp:581-660 */
size_t orc_refine_plane(const uint8_t* valid, const double* p3d, int w, int h,
                        const orc_refine_params* rp, double plane[4], double* moments)
{
    const int umin = rp->central_third_only ? w / 4 : 0;
    const int umax = rp->central_third_only ? w * 3 / 4 : w - 1;
    const int vmin = rp->central_third_only ? h / 4 : 0;
    const int vmax = rp->central_third_only ? h * 2 / 3 : h - 1;
    double c[3] = { 0, 0, 0 }, wsum = 0.0, A[9] = { 0, 0, 0, 0, 0, 0, 0, 0, 0 }, n[3], nn, d;
    size_t cnt = 0;
    int u, v, pass;
    for (pass = 0; pass < 2; pass++) {
        for (v = vmin; v <= vmax; ++v)
            for (u = umin; u <= umax; ++u) {
                const size_t i = (size_t)v * w + u;
                const double* p = p3d + i * 3;
                double dist, wt;
                if (!valid[i]) continue;
                dist = sqrt(p[0] * p[0] + p[1] * p[1] + p[2] * p[2]);
                if (!(p[0] > rp->xmin && p[0] < rp->xmax && p[1] > rp->ymin && p[1] < rp->ymax &&
                      dist < rp->max_distance))
                    continue;
                wt = rp->weight_by_distance ? dist : 1.0;
                if (pass == 0) {
                    cnt++;
                    wsum += wt;
                    c[0] += p[0] * wt; c[1] += p[1] * wt; c[2] += p[2] * wt;
                } else {
                    double q[3] = { p[0] - c[0], p[1] - c[1], p[2] - c[2] };
                    int a, b;
                    for (a = 0; a < 3; a++)
                        for (b = 0; b < 3; b++) A[a * 3 + b] = A[a * 3 + b] + wt * q[a] * q[b];
                }
            }
        if (pass == 0) { c[0] = c[0] / wsum; c[1] = c[1] / wsum; c[2] = c[2] / wsum; }
    }
    if (moments) { moments[0] = wsum; memcpy(moments + 1, c, sizeof c); memcpy(moments + 4, A, sizeof A); }
    orc_smallest_eigvec3(A, n);
    nn = sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
    n[0] /= nn; n[1] /= nn; n[2] /= nn;
    if (n[2] < 0) { n[0] *= -1.0; n[1] *= -1.0; n[2] *= -1.0; }
    d = -(n[0] * c[0] + n[1] * c[1] + n[2] * c[2]);
    plane[0] = n[0]; plane[1] = n[1]; plane[2] = n[2]; plane[3] = d;
    return cnt;
}

/* wass_stereo/PovMesh.cpp:1044-1069 */
void orc_RT_from_plane(const double plane[4], double R[9], double T[3],
                       double Rinv[9], double Tinv[3])
{
    const double a = plane[0], b = plane[1], c = plane[2], d = plane[3];
    const double q = (1 - c) / (a * a + b * b);
    int i, j;
    R[0] = 1 - a * a * q; R[1] = -a * b * q; R[2] = -a;
    R[3] = -a * b * q; R[4] = 1 - b * b * q; R[5] = -b;
    R[6] = a; R[7] = b; R[8] = c;
    for (i = 0; i < 3; i++) for (j = 0; j < 3; j++) Rinv[i * 3 + j] = R[j * 3 + i];
    T[0] = 0; T[1] = 0; T[2] = d;
    {
        double mT[3] = { -T[0], -T[1], -T[2] };
        mat3_mulv(Rinv, mT, Tinv);
    }
}

/* wass_stereo/PovMesh.cpp:377-460 */
size_t orc_encode_xyzc(const uint8_t* valid, const double* p3d, int w, int h,
                       const double plane[4], uint8_t* buf)
{
    const size_t N = (size_t)w * h;
    double R[9], T[3], Rinv[9], Tinv[3];
    double mn[3] = { 1.7976931348623157e308, 1.7976931348623157e308, 1.7976931348623157e308 };
    double mx[3] = { -1.7976931348623157e308, -1.7976931348623157e308, -1.7976931348623157e308 };
    double sc[3];
    uint32_t n = 0;
    size_t i, o;
    uint16_t* q;
    int k;
    orc_RT_from_plane(plane, R, T, Rinv, Tinv);
    for (i = 0; i < N; i++)
        if (valid[i]) {
            double pt[3];
            mat3_mulv(R, p3d + i * 3, pt);
            pt[0] += T[0]; pt[1] += T[1]; pt[2] += T[2];
            for (k = 0; k < 3; k++) { if (pt[k] < mn[k]) mn[k] = pt[k]; if (pt[k] > mx[k]) mx[k] = pt[k]; }
            n++;
        }
    for (k = 0; k < 3; k++) sc[k] = 65535.0 / (mx[k] - mn[k]);
    o = 0;
    memcpy(buf + o, &n, 4); o += 4;
    memcpy(buf + o, sc, 24); o += 24;
    memcpy(buf + o, mn, 24); o += 24;
    memcpy(buf + o, Rinv, 72); o += 72;
    memcpy(buf + o, Tinv, 24); o += 24;
    q = (uint16_t*)(buf + o);
    for (i = 0; i < N; i++)
        if (valid[i]) {
            double pt[3];
            mat3_mulv(R, p3d + i * 3, pt);
            pt[0] += T[0]; pt[1] += T[1]; pt[2] += T[2];
            for (k = 0; k < 3; k++) *q++ = (uint16_t)((pt[k] - mn[k]) * sc[k]);
        }
    return o + (size_t)n * 6;
}
