/*
 * rectify_oracle.c -- CPU restatement of rectify() (src/wass_stereo/wass_stereo.cpp:447-613), row f1.
 *
 * TEST INFRASTRUCTURE ONLY (see wass_oracle.h).
 *
 * PARITY UNPINNED: cv::stereoRectify, cv::initUndistortRectifyMap, cv::remap and cv::warpPerspective live in
 * OpenCV 4.5.5 (modules/calib3d/src/calibration.cpp, undistort.dispatch.cpp, modules/imgproc/src/imgwarp.cpp),
 * a third-party dependency that is neither vendored in /root/reference nor installed in this image; the
 * reference holds no golden rectified images.  The functions below restate the published algorithms and are
 * anchored on the call sites wass_stereo.cpp:515-516 (warpPerspective), :541 (stereoRectify), :600-601
 * (initUndistortRectifyMap), :603-604 (remap INTER_CUBIC).
 */
#include "wass_oracle.h"

/* gcc 11.4 -O3 drops the double->float->double round trips of the corner loop in orc_stereo_rectify when it
 * SLP-vectorises it (the result then differs from -O0/-O2 and from IEEE evaluation); keep that pass off here. */
#if defined(__GNUC__) && !defined(__clang__)
#pragma GCC optimize("no-tree-slp-vectorize")
#endif

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define ORC_INTER_BITS 5
#define ORC_TAB (1 << ORC_INTER_BITS)
#define ORC_COEF_SCALE 32768

static short sat_short_f(float v)
{
    long r = lrintf(v);
    return (short)(r < -32768 ? -32768 : (r > 32767 ? 32767 : r));
}

static void coeffs_1d(int ksize, float x, float* c)
{
    if (ksize == 2) {                                  /* interpolateLinear */
        c[0] = 1.f - x;
        c[1] = x;
    } else {                                           /* interpolateCubic, A = -0.75 */
        const float A = -0.75f;
        c[0] = ((A * (x + 1) - 5 * A) * (x + 1) + 8 * A) * (x + 1) - 4 * A;
        c[1] = ((A + 2) * x - (A + 3)) * x * x + 1;
        c[2] = ((A + 2) * (1 - x) - (A + 3)) * (1 - x) * (1 - x) + 1;
        c[3] = 1.f - c[0] - c[1] - c[2];
    }
}

/* initInterTab2D(method, fixpt=true): 32x32 entries of ksize x ksize int16 weights that sum to 2^15.
 * out holds 1024*ksize*ksize shorts.  The fix-up scan window [ksize/2, ksize/2+2) is OpenCV's; for the
 * bilinear table it looks past the entry into not-yet-written (zero) storage, reproduced with a zeroed tail. */
void orc_inter_tab(int ksize, int16_t* out)
{
    const int kk = ksize * ksize, n = ORC_TAB * ORC_TAB * kk;
    float t1[ORC_TAB * 4];
    short* itab = (short*)calloc((size_t)n + 64, sizeof(short));
    int i, j, k1, k2;
    for (i = 0; i < ORC_TAB; ++i) coeffs_1d(ksize, i * (1.f / ORC_TAB), t1 + i * ksize);
    for (i = 0; i < ORC_TAB; ++i)
        for (j = 0; j < ORC_TAB; ++j) {
            short* it = itab + (size_t)(i * ORC_TAB + j) * kk;
            int isum = 0;
            for (k1 = 0; k1 < ksize; ++k1) {
                float vy = t1[i * ksize + k1];
                for (k2 = 0; k2 < ksize; ++k2) {
                    float v = vy * t1[j * ksize + k2];
                    it[k1 * ksize + k2] = sat_short_f(v * ORC_COEF_SCALE);
                    isum += it[k1 * ksize + k2];
                }
            }
            if (isum != ORC_COEF_SCALE) {
                int diff = isum - ORC_COEF_SCALE;
                int ksize2 = ksize / 2, Mk1 = ksize2, Mk2 = ksize2, mk1 = ksize2, mk2 = ksize2;
                for (k1 = ksize2; k1 < ksize2 + 2; ++k1)
                    for (k2 = ksize2; k2 < ksize2 + 2; ++k2) {
                        if (it[k1 * ksize + k2] < it[mk1 * ksize + mk2]) { mk1 = k1; mk2 = k2; }
                        else if (it[k1 * ksize + k2] > it[Mk1 * ksize + Mk2]) { Mk1 = k1; Mk2 = k2; }
                    }
                if (diff < 0) it[Mk1 * ksize + Mk2] = (short)(it[Mk1 * ksize + Mk2] - diff);
                else it[mk1 * ksize + mk2] = (short)(it[mk1 * ksize + mk2] - diff);
            }
        }
    memcpy(out, itab, (size_t)n * sizeof(short));
    free(itab);
}

static uint8_t cast_u8(int v)                           /* FixedPtCast<int, uchar, 15> */
{
    v = (v + (1 << 14)) >> 15;
    return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
}
static int sat16(int v) { return v < -32768 ? -32768 : (v > 32767 ? 32767 : v); }
static int pix(const uint8_t* s, int sw, int sh, size_t ss, int y, int x)
{
    return (x >= 0 && x < sw && y >= 0 && y < sh) ? s[(size_t)y * ss + x] : 0;   /* BORDER_CONSTANT, value 0 */
}

/* remapBilinear / remapBicubic on one pixel with BORDER_CONSTANT 0: taps outside the source contribute the
 * border value, which is what both the fast interior branch and the border branch compute. */
static uint8_t tap_sum(const uint8_t* s, int sw, int sh, size_t ss, int sx, int sy, const int16_t* w, int ksize)
{
    int sum = 0, i, j;
    const int o = ksize / 2 - 1;
    for (i = 0; i < ksize; ++i)
        for (j = 0; j < ksize; ++j) sum += pix(s, sw, sh, ss, sy - o + i, sx - o + j) * w[i * ksize + j];
    return cast_u8(sum);
}

/* cv::warpPerspective(src, dst, H, Size(dw,dh)), INTER_LINEAR, BORDER_CONSTANT 0 (wass_stereo.cpp:515-516) */
void orc_warp_perspective(const uint8_t* src, int sw, int sh, size_t ss, const double H[9], int dw, int dh, uint8_t* dst)
{
    int16_t* tab = (int16_t*)malloc(sizeof(int16_t) * ORC_TAB * ORC_TAB * 4);
    double M[9], d;
    int x, y;
    const int bh0 = dh < 16 ? dh : 16;
    const int bw0 = (1024 / bh0) < dw ? (1024 / bh0) : dw;
    orc_inter_tab(2, tab);
    /* cv::invert, 3x3 closed form */
    d = H[0] * (H[4] * H[8] - H[5] * H[7]) - H[1] * (H[3] * H[8] - H[5] * H[6]) + H[2] * (H[3] * H[7] - H[4] * H[6]);
    d = 1. / d;
    M[0] = (H[4] * H[8] - H[5] * H[7]) * d; M[1] = (H[2] * H[7] - H[1] * H[8]) * d; M[2] = (H[1] * H[5] - H[2] * H[4]) * d;
    M[3] = (H[5] * H[6] - H[3] * H[8]) * d; M[4] = (H[0] * H[8] - H[2] * H[6]) * d; M[5] = (H[2] * H[3] - H[0] * H[5]) * d;
    M[6] = (H[3] * H[7] - H[4] * H[6]) * d; M[7] = (H[1] * H[6] - H[0] * H[7]) * d; M[8] = (H[0] * H[4] - H[1] * H[3]) * d;
    for (y = 0; y < dh; ++y)
        for (x = 0; x < dw; ++x) {
            const int bx = (x / bw0) * bw0, x1 = x - bx;
            double X0 = M[0] * bx + M[1] * y + M[2];
            double Y0 = M[3] * bx + M[4] * y + M[5];
            double W0 = M[6] * bx + M[7] * y + M[8];
            double W = W0 + M[6] * x1, fX, fY;
            int X, Y, a;
            W = W ? 32. / W : 0;
            fX = fmax(-2147483648.0, fmin(2147483647.0, (X0 + M[0] * x1) * W));
            fY = fmax(-2147483648.0, fmin(2147483647.0, (Y0 + M[3] * x1) * W));
            X = (int)lrint(fX);
            Y = (int)lrint(fY);
            a = (Y & 31) * 32 + (X & 31);
            dst[(size_t)y * dw + x] = tap_sum(src, sw, sh, ss, sat16(X >> 5), sat16(Y >> 5), tab + a * 4, 2);
        }
    free(tab);
}

/* cv::remap(src, dst, map1, map2, INTER_CUBIC), CV_32FC1 maps, BORDER_CONSTANT 0 (wass_stereo.cpp:603-604) */
void orc_remap_cubic(const uint8_t* src, int sw, int sh, size_t ss, const float* mx, const float* my, int dw, int dh, uint8_t* dst)
{
    int16_t* tab = (int16_t*)malloc(sizeof(int16_t) * ORC_TAB * ORC_TAB * 16);
    size_t i, n = (size_t)dw * dh;
    orc_inter_tab(4, tab);
    for (i = 0; i < n; ++i) {
        const int X = (int)lrintf(mx[i] * 32.f), Y = (int)lrintf(my[i] * 32.f);
        const int a = (Y & 31) * 32 + (X & 31);
        dst[i] = tap_sum(src, sw, sh, ss, sat16(X >> 5), sat16(Y >> 5), tab + a * 16, 4);
    }
    free(tab);
}

/* ---- small 3x3 helpers ---- */
static void m33(const double* a, const double* b, double* o)
{
    double t[9];
    int i, j;
    for (i = 0; i < 3; ++i)
        for (j = 0; j < 3; ++j) t[i * 3 + j] = a[i * 3] * b[j] + a[i * 3 + 1] * b[3 + j] + a[i * 3 + 2] * b[6 + j];
    memcpy(o, t, sizeof t);
}
static void m33t(const double* a, const double* b, double* o)
{
    double t[9];
    int i, j;
    for (i = 0; i < 3; ++i)
        for (j = 0; j < 3; ++j) t[i * 3 + j] = a[i * 3] * b[j * 3] + a[i * 3 + 1] * b[j * 3 + 1] + a[i * 3 + 2] * b[j * 3 + 2];
    memcpy(o, t, sizeof t);
}
static void m3v(const double* a, const double* v, double* o)
{
    double t[3];
    int i;
    for (i = 0; i < 3; ++i) t[i] = a[i * 3] * v[0] + a[i * 3 + 1] * v[1] + a[i * 3 + 2] * v[2];
    memcpy(o, t, sizeof t);
}
static int inv3(const double* m, double* o)
{
    double t[9];
    double d = m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) + m[2] * (m[3] * m[7] - m[4] * m[6]);
    if (d == 0) return 0;
    d = 1. / d;
    t[0] = (m[4] * m[8] - m[5] * m[7]) * d; t[1] = (m[2] * m[7] - m[1] * m[8]) * d; t[2] = (m[1] * m[5] - m[2] * m[4]) * d;
    t[3] = (m[5] * m[6] - m[3] * m[8]) * d; t[4] = (m[0] * m[8] - m[2] * m[6]) * d; t[5] = (m[2] * m[3] - m[0] * m[5]) * d;
    t[6] = (m[3] * m[7] - m[4] * m[6]) * d; t[7] = (m[1] * m[6] - m[0] * m[7]) * d; t[8] = (m[0] * m[4] - m[1] * m[3]) * d;
    memcpy(o, t, sizeof t);
    return 1;
}

/* cv::initUndistortRectifyMap(K, zeros, R, P, size, CV_32FC1) (wass_stereo.cpp:600-601) */
int orc_init_rectify_map(const double K[9], const double R[9], const double P[12], int w, int h, float* mx, float* my)
{
    double P33[9], PR[9], ir[9];
    int i, j;
    for (i = 0; i < 3; ++i)
        for (j = 0; j < 3; ++j) P33[i * 3 + j] = P[i * 4 + j];
    m33(P33, R, PR);
    if (!inv3(PR, ir)) return -1;
    for (i = 0; i < h; ++i) {
        double _x = i * ir[1] + ir[2], _y = i * ir[4] + ir[5], _w = i * ir[7] + ir[8];
        for (j = 0; j < w; ++j, _x += ir[0], _y += ir[3], _w += ir[6]) {
            double ww = 1. / _w, x = _x * ww, y = _y * ww;
            /* zero distortion: kr = 1, xd = x, yd = y, identity tilt */
            mx[(size_t)i * w + j] = (float)(K[0] * x + K[2]);
            my[(size_t)i * w + j] = (float)(K[4] * y + K[5]);
        }
    }
    return 0;
}

/* cvRodrigues2 */
static void rod_v2m(const double* r, double* R)
{
    double theta = sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
    int i;
    if (theta < DBL_EPSILON) {
        for (i = 0; i < 9; ++i) R[i] = (i % 4 == 0) ? 1. : 0.;
        return;
    }
    {
        double c = cos(theta), s = sin(theta), c1 = 1. - c, it = 1. / theta;
        double x = r[0] * it, y = r[1] * it, z = r[2] * it;
        double rrt[9], rx[9];
        rrt[0] = x * x; rrt[1] = x * y; rrt[2] = x * z; rrt[3] = x * y; rrt[4] = y * y; rrt[5] = y * z; rrt[6] = x * z; rrt[7] = y * z; rrt[8] = z * z;
        rx[0] = 0; rx[1] = -z; rx[2] = y; rx[3] = z; rx[4] = 0; rx[5] = -x; rx[6] = -y; rx[7] = x; rx[8] = 0;
        for (i = 0; i < 9; ++i) R[i] = c * ((i % 4 == 0) ? 1. : 0.) + c1 * rrt[i] + s * rx[i];
    }
}
static void rod_m2v(const double* R, double* r)
{
    double rx = R[7] - R[5], ry = R[2] - R[6], rz = R[3] - R[1];
    double s = sqrt((rx * rx + ry * ry + rz * rz) * 0.25);
    double c = (R[0] + R[4] + R[8] - 1) * 0.5, theta;
    c = c > 1. ? 1. : (c < -1. ? -1. : c);
    theta = acos(c);
    if (s < 1e-5) {
        double t;
        if (c > 0) { r[0] = r[1] = r[2] = 0; return; }
        t = (R[0] + 1) * 0.5; rx = sqrt(t > 0 ? t : 0);
        t = (R[4] + 1) * 0.5; ry = sqrt(t > 0 ? t : 0) * (R[1] < 0 ? -1. : 1.);
        t = (R[8] + 1) * 0.5; rz = sqrt(t > 0 ? t : 0) * (R[2] < 0 ? -1. : 1.);
        if (fabs(rx) < fabs(ry) && fabs(rx) < fabs(rz) && (R[5] > 0) != (ry * rz > 0)) rz = -rz;
        theta /= sqrt(rx * rx + ry * ry + rz * rz);
        r[0] = rx * theta; r[1] = ry * theta; r[2] = rz * theta;
        return;
    }
    {
        double vth = 1 / (2 * s);
        vth *= theta;
        r[0] = rx * vth; r[1] = ry * vth; r[2] = rz * vth;
    }
}

typedef struct { float x, y, width, height; } rectf;

/* icvGetRectangles: undistortPoints of a 9x9 grid through R and the new camera matrix */
static void get_rects(const double* K, const double* R, const double* P, int W, int H, rectf* inner, rectf* outer)
{
    const int N = 9;
    double P33[9], RR[9];
    float iX0 = -FLT_MAX, iX1 = FLT_MAX, iY0 = -FLT_MAX, iY1 = FLT_MAX, oX0 = FLT_MAX, oX1 = -FLT_MAX, oY0 = FLT_MAX, oY1 = -FLT_MAX;
    int x, y, i, j;
    for (i = 0; i < 3; ++i)
        for (j = 0; j < 3; ++j) P33[i * 3 + j] = P[i * 4 + j];
    m33(P33, R, RR);
    for (y = 0; y < N; ++y)
        for (x = 0; x < N; ++x) {
            float fx = (float)x * (W - 1) / (N - 1), fy = (float)y * (H - 1) / (N - 1);
            double xn = (fx - K[2]) * (1. / K[0]), yn = (fy - K[5]) * (1. / K[4]);
            double xx = RR[0] * xn + RR[1] * yn + RR[2], yy = RR[3] * xn + RR[4] * yn + RR[5], ww = 1. / (RR[6] * xn + RR[7] * yn + RR[8]);
            float px = (float)(xx * ww), py = (float)(yy * ww);
            if (px < oX0) oX0 = px;
            if (px > oX1) oX1 = px;
            if (py < oY0) oY0 = py;
            if (py > oY1) oY1 = py;
            if (x == 0 && px > iX0) iX0 = px;
            if (x == N - 1 && px < iX1) iX1 = px;
            if (y == 0 && py > iY0) iY0 = py;
            if (y == N - 1 && py < iY1) iY1 = py;
        }
    inner->x = iX0; inner->y = iY0; inner->width = iX1 - iX0; inner->height = iY1 - iY0;
    outer->x = oX0; outer->y = oY0; outer->width = oX1 - oX0; outer->height = oY1 - oY0;
}

static double max2(double a, double b) { return a > b ? a : b; }
static double min2(double a, double b) { return a < b ? a : b; }

static void clip_roi(double x, double y, double w, double h, int W, int H, int* roi)
{
    int rx = (int)ceil(x), ry = (int)ceil(y), rw = (int)floor(w), rh = (int)floor(h);
    int x1 = rx > 0 ? rx : 0, y1 = ry > 0 ? ry : 0;
    int w1 = (rx + rw < W ? rx + rw : W) - x1, h1 = (ry + rh < H ? ry + rh : H) - y1;
    if (w1 <= 0 || h1 <= 0) { roi[0] = roi[1] = roi[2] = roi[3] = 0; return; }
    roi[0] = x1; roi[1] = y1; roi[2] = w1; roi[3] = h1;
}

/* cv::stereoRectify(K1, 0, K2, 0, size, R, T, ..., flags=0, alpha, size, &roi1, &roi2) (wass_stereo.cpp:541) */
int orc_stereo_rectify(const double K1[9], const double K2[9], int W, int H, const double R[9], const double T[3], double alpha,
                       double R1[9], double R2[9], double P1[12], double P2[12], int roi1[4], int roi2[4])
{
    double om[3], r_r[9], t[3], uu[3] = { 0, 0, 0 }, ww[3], wR[9], c, nt, nw, fc_new, cc[2][2], s = 1.;
    rectf in1, out1, in2, out2;
    int idx, i, k;
    rod_m2v(R, om);
    for (i = 0; i < 3; ++i) om[i] *= -0.5;
    rod_v2m(om, r_r);
    m3v(r_r, T, t);
    idx = fabs(t[0]) > fabs(t[1]) ? 0 : 1;
    c = t[idx];
    nt = sqrt(t[0] * t[0] + t[1] * t[1] + t[2] * t[2]);
    if (!(nt > 0.0)) return -1;
    uu[idx] = c > 0 ? 1 : -1;
    ww[0] = t[1] * uu[2] - t[2] * uu[1];
    ww[1] = t[2] * uu[0] - t[0] * uu[2];
    ww[2] = t[0] * uu[1] - t[1] * uu[0];
    nw = sqrt(ww[0] * ww[0] + ww[1] * ww[1] + ww[2] * ww[2]);
    if (nw > 0.0) {
        double sc = acos(fabs(c) / nt) / nw;
        for (i = 0; i < 3; ++i) ww[i] *= sc;
    }
    rod_v2m(ww, wR);
    m33t(wR, r_r, R1);
    m33(wR, r_r, R2);
    m3v(R2, T, t);
    fc_new = (K1[(idx ^ 1) * 4] + K2[(idx ^ 1) * 4]) * 0.5;
    for (k = 0; k < 2; ++k) {
        const double* A = k == 0 ? K1 : K2;
        const double* Rk = k == 0 ? R1 : R2;
        double ax = 0, ay = 0;
        for (i = 0; i < 4; ++i) {
            int j = i < 2 ? 0 : 1;
            float px = (float)((i % 2) * (W - 1)), py = (float)(j * (H - 1));
            float xn = (float)((px - A[2]) * (1. / A[0])), yn = (float)((py - A[5]) * (1. / A[4]));
            double X = xn, Y = yn;
            double x = Rk[0] * X + Rk[1] * Y + Rk[2], y = Rk[3] * X + Rk[4] * Y + Rk[5], z = Rk[6] * X + Rk[7] * Y + Rk[8];
            z = z ? 1. / z : 1;
            ax += (double)(float)(x * z * fc_new + 0.0);
            ay += (double)(float)(y * z * fc_new + 0.0);
        }
        cc[k][0] = (W - 1) / 2 - ax / 4;
        cc[k][1] = (H - 1) / 2 - ay / 4;
    }
    if (idx == 0) cc[0][1] = cc[1][1] = (cc[0][1] + cc[1][1]) * 0.5;
    else cc[0][0] = cc[1][0] = (cc[0][0] + cc[1][0]) * 0.5;
    memset(P1, 0, 12 * sizeof(double));
    memset(P2, 0, 12 * sizeof(double));
    P1[0] = P1[5] = fc_new; P1[2] = cc[0][0]; P1[6] = cc[0][1]; P1[10] = 1;
    P2[0] = P2[5] = fc_new; P2[2] = cc[1][0]; P2[6] = cc[1][1]; P2[10] = 1;
    P2[idx * 4 + 3] = t[idx] * fc_new;
    alpha = alpha < 1. ? alpha : 1.;
    get_rects(K1, R1, P1, W, H, &in1, &out1);
    get_rects(K2, R2, P2, W, H, &in2, &out2);
    {
        double cx1_0 = cc[0][0], cy1_0 = cc[0][1], cx2_0 = cc[1][0], cy2_0 = cc[1][1];
        double cx1 = W * cx1_0 / W, cy1 = H * cy1_0 / H, cx2 = W * cx2_0 / W, cy2 = H * cy2_0 / H;
        if (alpha >= 0) {
            double s0 = max2(max2(max2(cx1 / (cx1_0 - in1.x), cy1 / (cy1_0 - in1.y)), (W - cx1) / (in1.x + in1.width - cx1_0)),
                             (H - cy1) / (in1.y + in1.height - cy1_0));
            double s1 = min2(min2(min2(cx1 / (cx1_0 - out1.x), cy1 / (cy1_0 - out1.y)), (W - cx1) / (out1.x + out1.width - cx1_0)),
                             (H - cy1) / (out1.y + out1.height - cy1_0));
            s0 = max2(max2(max2(max2(cx2 / (cx2_0 - in2.x), cy2 / (cy2_0 - in2.y)), (W - cx2) / (in2.x + in2.width - cx2_0)),
                           (H - cy2) / (in2.y + in2.height - cy2_0)), s0);
            s1 = min2(min2(min2(min2(cx2 / (cx2_0 - out2.x), cy2 / (cy2_0 - out2.y)), (W - cx2) / (out2.x + out2.width - cx2_0)),
                           (H - cy2) / (out2.y + out2.height - cy2_0)), s1);
            s = s0 * (1 - alpha) + s1 * alpha;
        }
        fc_new *= s;
        P1[0] = P1[5] = fc_new; P1[2] = cx1; P1[6] = cy1;
        P2[0] = P2[5] = fc_new; P2[2] = cx2; P2[6] = cy2;
        P2[idx * 4 + 3] = s * P2[idx * 4 + 3];
        if (roi1) clip_roi((in1.x - cx1_0) * s + cx1, (in1.y - cy1_0) * s + cy1, in1.width * s, in1.height * s, W, H, roi1);
        if (roi2) clip_roi((in2.x - cx2_0) * s + cx2, (in2.y - cy2_0) * s + cy2, in2.width * s, in2.height * s, W, H, roi2);
    }
    return 0;
}

/* cv::undistort(src, dst, K, dist) as called by wass_prepare (src/wass_prepare/wass_prepare.cpp:268): new camera
 * matrix = K, INTER_LINEAR, BORDER_CONSTANT 0.  Restated from OpenCV 4.5.5 imgproc/undistort.dispatch.cpp:
 * the image is processed in stripes of max(1, 4096 / cols) rows; each stripe builds CV_16SC2 + CV_16UC1
 * fixed-point maps with initUndistortRectifyMap (R = I, principal point shifted by the stripe origin) and remaps.
 * dist holds n = 4, 5, 8 or 12 coefficients (k1 k2 p1 p2 [k3 [k4 k5 k6 [s1 s2 s3 s4]]]); the tilt model is not
 * restated. */
int orc_undistort(const uint8_t* src, int w, int h, size_t ss, const double K[9], const double* dist, int n, uint8_t* dst)
{
    double k[12] = { 0 }, A[9], Ar[9], ir[9];
    const double fx = K[0], fy = K[4], u0 = K[2], v0 = K[5];
    int16_t* tab;
    int stripe0, y0, i, j;
    if (!(n == 4 || n == 5 || n == 8 || n == 12)) return -1;
    for (i = 0; i < n; ++i) k[i] = dist[i];
    {
        /* OpenCV's order: k1 k2 p1 p2 k3 k4 k5 k6 s1 s2 s3 s4 */
    }
    memcpy(A, K, sizeof A);
    tab = (int16_t*)malloc(sizeof(int16_t) * ORC_TAB * ORC_TAB * 4);
    orc_inter_tab(2, tab);
    stripe0 = 4096 / (w > 1 ? w : 1);
    if (stripe0 < 1) stripe0 = 1;
    if (stripe0 > h) stripe0 = h;
    for (y0 = 0; y0 < h; y0 += stripe0) {
        const int stripe = stripe0 < h - y0 ? stripe0 : h - y0;
        memcpy(Ar, A, sizeof Ar);
        Ar[5] = A[5] - y0;
        if (!inv3(Ar, ir)) { free(tab); return -1; }
        for (i = 0; i < stripe; ++i) {
            double _x = i * ir[1] + ir[2], _y = i * ir[4] + ir[5], _w = i * ir[7] + ir[8];
            for (j = 0; j < w; ++j, _x += ir[0], _y += ir[3], _w += ir[6]) {
                double ww = 1. / _w, x = _x * ww, y = _y * ww;
                double x2 = x * x, y2 = y * y;
                double r2 = x2 + y2, _2xy = 2 * x * y;
                double kr = (1 + ((k[4] * r2 + k[1]) * r2 + k[0]) * r2) / (1 + ((k[7] * r2 + k[6]) * r2 + k[5]) * r2);
                double xd = (x * kr + k[2] * _2xy + k[3] * (r2 + 2 * x2) + k[8] * r2 + k[9] * r2 * r2);
                double yd = (y * kr + k[2] * (r2 + 2 * y2) + k[3] * _2xy + k[10] * r2 + k[11] * r2 * r2);
                double u = fx * xd + u0, v = fy * yd + v0;               /* identity tilt, invProj = 1 */
                double fu = u * 32, fv = v * 32;
                int iu, iv, a;
                fu = fu < -2147483648.0 ? -2147483648.0 : (fu > 2147483647.0 ? 2147483647.0 : fu);
                fv = fv < -2147483648.0 ? -2147483648.0 : (fv > 2147483647.0 ? 2147483647.0 : fv);
                iu = (int)lrint(fu);
                iv = (int)lrint(fv);
                a = (iv & 31) * 32 + (iu & 31);
                dst[(size_t)(y0 + i) * w + j] = tap_sum(src, w, h, ss, (int16_t)(iu >> 5), (int16_t)(iv >> 5), tab + a * 4, 2);
            }
        }
    }
    free(tab);
    return 0;
}
