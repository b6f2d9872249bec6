// rectify.hip -- row f1 of SURVEY.md section 8: rectify() of wass_stereo.cpp:447-613.
//
//   rig-constant host math      cv::stereoRectify (:541), cv::initUndistortRectifyMap (:600-601)
//   per-frame GPU resamplers    cv::remap INTER_CUBIC (:603-604), cv::warpPerspective INTER_LINEAR (:515-516)
//
// OpenCV is a third-party dependency that is absent from this image, so everything here restates OpenCV 4.5.5's
// published algorithms (calib3d/calibration.cpp, calib3d/undistort.dispatch.cpp, imgproc/imgwarp.cpp) from
// knowledge: PARITY UNPINNED against OpenCV, bit-exact against oracle/rectify_oracle.c (same restatement).
// The resamplers reproduce OpenCV's integer pipeline: source coordinates quantised to 1/32 pixel, weights taken
// from 32x32 tables of int16 products scaled by 2^15 (with OpenCV's sum fix-up), (sum + 2^14) >> 15, constant-0
// border taps.
#include "common.h"

#include <cfloat>
#include <cmath>
#include <cstring>
#include <vector>

namespace wass {

constexpr int INTER_BITS = 5, TAB = 1 << INTER_BITS, TAB2 = TAB * TAB, COEF_SCALE = 1 << 15;

// ---------------------------------------------------------------- interpolation tables (imgwarp.cpp initInterTab2D)
static short sat_short(float v)
{
    const long r = lrintf(v);      // cvRound: nearest even
    return (short)(r < -32768 ? -32768 : r > 32767 ? 32767 : r);
}

static void tab1d(int ksize, float x, float* c)
{
    if (ksize == 2) { c[0] = 1.f - x; c[1] = x; return; }
    const float A = -0.75f;
    c[0] = ((A * (x + 1) - 5 * A) * (x + 1) + 8 * A) * (x + 1) - 4 * A;
    c[1] = ((A + 2) * x - (A + 3)) * x * x + 1;
    c[2] = ((A + 2) * (1 - x) - (A + 3)) * (1 - x) * (1 - x) + 1;
    c[3] = 1.f - c[0] - c[1] - c[2];
}

// itab[(fy*32+fx)*ksize*ksize + ky*ksize + kx]
static void build_itab(int ksize, std::vector<short>& itab)
{
    std::vector<float> t1((size_t)TAB * ksize);
    const float scale = 1.f / TAB;
    for (int i = 0; i < TAB; ++i) tab1d(ksize, i * scale, &t1[(size_t)i * ksize]);
    const int kk = ksize * ksize;
    itab.assign((size_t)TAB2 * kk + 64, 0);        // zero slack: the fix-up scan below may look past an entry
    for (int i = 0; i < TAB; ++i)
        for (int j = 0; j < TAB; ++j) {
            short* it = &itab[(size_t)(i * TAB + j) * kk];
            int isum = 0;
            for (int k1 = 0; k1 < ksize; ++k1) {
                const float vy = t1[(size_t)i * ksize + k1];
                for (int k2 = 0; k2 < ksize; ++k2) {
                    const float v = vy * t1[(size_t)j * ksize + k2];
                    isum += it[k1 * ksize + k2] = sat_short(v * COEF_SCALE);
                }
            }
            if (isum != COEF_SCALE) {
                const int diff = isum - COEF_SCALE;
                const int k0 = ksize / 2;
                int Mk1 = k0, Mk2 = k0, mk1 = k0, mk2 = k0;
                for (int k1 = k0; k1 < k0 + 2; ++k1)
                    for (int k2 = k0; k2 < k0 + 2; ++k2) {
                        if (it[k1 * ksize + k2] < it[mk1 * ksize + mk2]) { mk1 = k1; mk2 = k2; }
                        else if (it[k1 * ksize + k2] > it[Mk1 * ksize + Mk2]) { Mk1 = k1; Mk2 = k2; }
                    }
                if (diff < 0) it[Mk1 * ksize + Mk2] = (short)(it[Mk1 * ksize + Mk2] - diff);
                else it[mk1 * ksize + mk2] = (short)(it[mk1 * ksize + mk2] - diff);
            }
        }
    itab.resize((size_t)TAB2 * kk);
}

constexpr size_t TAB_LIN_OFF = 0, TAB_CUB_OFF = (size_t)TAB2 * 4;          // in shorts
constexpr size_t TAB_SHORTS = (size_t)TAB2 * 4 + (size_t)TAB2 * 16;

static int ensure_tables(wass_ctx* c)
{
    if (c->rect_tab_ready) return WASS_OK;
    int rc = ensure(c, c->rect_tab, TAB_SHORTS * sizeof(short));
    if (rc) return rc;
    std::vector<short> lin, cub, all;
    build_itab(2, lin);
    build_itab(4, cub);
    all.insert(all.end(), lin.begin(), lin.end());
    all.insert(all.end(), cub.begin(), cub.end());
    WASS_HIP(c, hipMemcpyAsync(c->rect_tab.p, all.data(), TAB_SHORTS * sizeof(short), hipMemcpyHostToDevice, c->stream));
    WASS_HIP(c, hipStreamSynchronize(c->stream));       // `all` is pageable host memory about to go out of scope
    c->rect_tab_ready = true;
    return WASS_OK;
}

// ---------------------------------------------------------------- kernels
__device__ __forceinline__ uint8_t fixed_cast_u8(int v)
{
    v = (v + (1 << 14)) >> 15;
    return (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
}
__device__ __forceinline__ int sat_s16(int v) { return v < -32768 ? -32768 : v > 32767 ? 32767 : v; }

// remapBilinear (imgwarp.cpp), BORDER_CONSTANT 0
__device__ __forceinline__ uint8_t sample_linear(const uint8_t* __restrict__ src, int sw, int sh, size_t ss, int sx, int sy,
                                                 const short* __restrict__ w)
{
    if ((unsigned)sx < (unsigned)max(sw - 1, 0) && (unsigned)sy < (unsigned)max(sh - 1, 0)) {
        const uint8_t* s0 = src + (size_t)sy * ss + sx;
        return fixed_cast_u8(s0[0] * w[0] + s0[1] * w[1] + s0[ss] * w[2] + s0[ss + 1] * w[3]);
    }
    if (sx >= sw || sx + 1 < 0 || sy >= sh || sy + 1 < 0) return 0;
    auto at = [&](int yy, int xx) -> int { return ((unsigned)xx < (unsigned)sw && (unsigned)yy < (unsigned)sh) ? src[(size_t)yy * ss + xx] : 0; };
    return fixed_cast_u8(at(sy, sx) * w[0] + at(sy, sx + 1) * w[1] + at(sy + 1, sx) * w[2] + at(sy + 1, sx + 1) * w[3]);
}

// remapBicubic (imgwarp.cpp), BORDER_CONSTANT 0; (sx, sy) is the integer source position (tap 1 of 4)
__device__ __forceinline__ uint8_t sample_cubic(const uint8_t* __restrict__ src, int sw, int sh, size_t ss, int sx, int sy,
                                                const short* __restrict__ w)
{
    sx -= 1; sy -= 1;
    int sum = 0;
    if ((unsigned)sx < (unsigned)max(sw - 3, 0) && (unsigned)sy < (unsigned)max(sh - 3, 0)) {
        const uint8_t* s = src + (size_t)sy * ss + sx;
#pragma unroll
        for (int i = 0; i < 4; ++i, s += ss)
            sum += s[0] * w[i * 4] + s[1] * w[i * 4 + 1] + s[2] * w[i * 4 + 2] + s[3] * w[i * 4 + 3];
        return fixed_cast_u8(sum);
    }
    if (sx >= sw || sx + 4 <= 0 || sy >= sh || sy + 4 <= 0) return 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int yy = sy + i;
        if ((unsigned)yy >= (unsigned)sh) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int xx = sx + j;
            if ((unsigned)xx < (unsigned)sw) sum += src[(size_t)yy * ss + xx] * w[i * 4 + j];
        }
    }
    return fixed_cast_u8(sum);
}

struct M9 { double m[9]; };

// WarpPerspectiveInvoker (imgwarp.cpp): coordinates are evaluated per block of bw0 destination columns as
// (X0(block start) + M0*x1) * (32 / (W0 + M6*x1)), rounded to nearest even.
__global__ __launch_bounds__(256) void k_warp_linear(const uint8_t* __restrict__ src, int sw, int sh, size_t ss, M9 M, int bw0,
                                                     int ox, int oy, int ow, int oh, uint8_t* __restrict__ dst,
                                                     const short* __restrict__ tab)
{
    const int tx = blockIdx.x * 64 + threadIdx.x, ty = blockIdx.y * 4 + threadIdx.y;
    if (tx >= ow || ty >= oh) return;
    const int x = ox + tx, y = oy + ty;
    const int bx = (x / bw0) * bw0, x1 = x - bx;
    const double* m = M.m;
    const double X0 = m[0] * bx + m[1] * y + m[2];
    const double Y0 = m[3] * bx + m[4] * y + m[5];
    const double W0 = m[6] * bx + m[7] * y + m[8];
    double W = W0 + m[6] * x1;
    W = W != 0.0 ? 32.0 / W : 0.0;
    const double fX = fmax(-2147483648.0, fmin(2147483647.0, (X0 + m[0] * x1) * W));
    const double fY = fmax(-2147483648.0, fmin(2147483647.0, (Y0 + m[3] * x1) * W));
    const int X = __double2int_rn(fX), Y = __double2int_rn(fY);
    const int sx = sat_s16(X >> INTER_BITS), sy = sat_s16(Y >> INTER_BITS);
    const int a = (Y & (TAB - 1)) * TAB + (X & (TAB - 1));
    dst[(size_t)ty * ow + tx] = sample_linear(src, sw, sh, ss, sx, sy, tab + TAB_LIN_OFF + (size_t)a * 4);
}

// cv::remap with two CV_32FC1 maps: sx = cvRound(map*32) (float product, nearest even)
__global__ __launch_bounds__(256) void k_remap_cubic(const uint8_t* __restrict__ src, int sw, int sh, size_t ss,
                                                     const float* __restrict__ mx, const float* __restrict__ my, int dw,
                                                     int ox, int oy, int ow, int oh, uint8_t* __restrict__ dst,
                                                     const short* __restrict__ tab)
{
    const int tx = blockIdx.x * 64 + threadIdx.x, ty = blockIdx.y * 4 + threadIdx.y;
    if (tx >= ow || ty >= oh) return;
    const size_t mi = (size_t)(oy + ty) * dw + (ox + tx);
    const int X = __float2int_rn(mx[mi] * (float)TAB), Y = __float2int_rn(my[mi] * (float)TAB);
    const int sx = sat_s16(X >> INTER_BITS), sy = sat_s16(Y >> INTER_BITS);
    const int a = (Y & (TAB - 1)) * TAB + (X & (TAB - 1));
    dst[(size_t)ty * ow + tx] = sample_cubic(src, sw, sh, ss, sx, sy, tab + TAB_CUB_OFF + (size_t)a * 16);
}

// cv::undistort: normalised coordinates per column / per row come from the host (the column sequence is an
// accumulated sum in OpenCV, the row value depends on the stripe the row belongs to); the distortion polynomial,
// the 1/32-pixel quantisation and the bilinear taps run here, one thread per pixel.
struct Dist12 { double k[12]; };
__global__ __launch_bounds__(256) void k_undistort(const uint8_t* __restrict__ src, int w, int h, size_t ss,
                                                   const double* __restrict__ xs, const double* __restrict__ ys, Dist12 D,
                                                   double fx, double fy, double u0, double v0, uint8_t* __restrict__ dst,
                                                   const short* __restrict__ tab)
{
    const int tx = blockIdx.x * 64 + threadIdx.x, ty = blockIdx.y * 4 + threadIdx.y;
    if (tx >= w || ty >= h) return;
    const double* k = D.k;
    const double x = xs[tx], y = ys[ty];
    const double x2 = x * x, y2 = y * y;
    const double r2 = x2 + y2, _2xy = 2 * x * y;
    const double kr = (1 + ((k[4] * r2 + k[1]) * r2 + k[0]) * r2) / (1 + ((k[7] * r2 + k[6]) * r2 + k[5]) * r2);
    const double xd = (x * kr + k[2] * _2xy + k[3] * (r2 + 2 * x2) + k[8] * r2 + k[9] * r2 * r2);
    const double yd = (y * kr + k[2] * (r2 + 2 * y2) + k[3] * _2xy + k[10] * r2 + k[11] * r2 * r2);
    const double u = fx * xd + u0, v = fy * yd + v0;
    const int iu = __double2int_rn(fmax(-2147483648.0, fmin(2147483647.0, u * 32)));
    const int iv = __double2int_rn(fmax(-2147483648.0, fmin(2147483647.0, v * 32)));
    const int a = (iv & (TAB - 1)) * TAB + (iu & (TAB - 1));
    dst[(size_t)ty * w + tx] = sample_linear(src, w, h, ss, (int)(short)(iu >> INTER_BITS), (int)(short)(iv >> INTER_BITS),
                                             tab + TAB_LIN_OFF + (size_t)a * 4);
}

// ---------------------------------------------------------------- small host linear algebra (row-major 3x3)
static void mul33(const double* a, const double* b, double* o)
{
    double t[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) t[i * 3 + j] = a[i * 3] * b[j] + a[i * 3 + 1] * b[3 + j] + a[i * 3 + 2] * b[6 + j];
    memcpy(o, t, sizeof t);
}
static void mul33T(const double* a, const double* b, double* o)          // a * b^T
{
    double t[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) t[i * 3 + j] = a[i * 3] * b[j * 3] + a[i * 3 + 1] * b[j * 3 + 1] + a[i * 3 + 2] * b[j * 3 + 2];
    memcpy(o, t, sizeof t);
}
static void mulv(const double* a, const double* v, double* o)
{
    double t[3];
    for (int i = 0; i < 3; ++i) t[i] = a[i * 3] * v[0] + a[i * 3 + 1] * v[1] + a[i * 3 + 2] * v[2];
    memcpy(o, t, sizeof t);
}
// cv::invert of a 3x3 (closed form used for n <= 3)
static bool inv33(const double* m, double* t)
{
    double d = m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) + m[2] * (m[3] * m[7] - m[4] * m[6]);
    if (d == 0) return false;
    d = 1. / d;
    double r[9];
    r[0] = (m[4] * m[8] - m[5] * m[7]) * d; r[1] = (m[2] * m[7] - m[1] * m[8]) * d; r[2] = (m[1] * m[5] - m[2] * m[4]) * d;
    r[3] = (m[5] * m[6] - m[3] * m[8]) * d; r[4] = (m[0] * m[8] - m[2] * m[6]) * d; r[5] = (m[2] * m[3] - m[0] * m[5]) * d;
    r[6] = (m[3] * m[7] - m[4] * m[6]) * d; r[7] = (m[1] * m[6] - m[0] * m[7]) * d; r[8] = (m[0] * m[4] - m[1] * m[3]) * d;
    memcpy(t, r, sizeof r);
    return true;
}

// cvRodrigues2, vector -> matrix
static void rodrigues_v2m(const double* r, double* R)
{
    const double theta = std::sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
    if (theta < DBL_EPSILON) { for (int i = 0; i < 9; ++i) R[i] = (i % 4 == 0) ? 1.0 : 0.0; return; }
    const double c = std::cos(theta), s = std::sin(theta), c1 = 1. - c, it = 1. / theta;
    const double x = r[0] * it, y = r[1] * it, z = r[2] * it;
    const double rrt[9] = { x * x, x * y, x * z, x * y, y * y, y * z, x * z, y * z, z * z };
    const double rx[9] = { 0, -z, y, z, 0, -x, -y, x, 0 };
    for (int i = 0; i < 9; ++i) R[i] = c * ((i % 4 == 0) ? 1.0 : 0.0) + c1 * rrt[i] + s * rx[i];
}
// cvRodrigues2, matrix -> vector (without the SVD re-orthogonalisation of the input: R is a rotation already)
static void rodrigues_m2v(const double* R, double* r)
{
    double rx = R[7] - R[5], ry = R[2] - R[6], rz = R[3] - R[1];
    const double s = std::sqrt((rx * rx + ry * ry + rz * rz) * 0.25);
    double c = (R[0] + R[4] + R[8] - 1) * 0.5;
    c = c > 1. ? 1. : (c < -1. ? -1. : c);
    double theta = std::acos(c);
    if (s < 1e-5) {
        if (c > 0) { r[0] = r[1] = r[2] = 0; return; }
        double t;
        t = (R[0] + 1) * 0.5; rx = std::sqrt(t > 0 ? t : 0);
        t = (R[4] + 1) * 0.5; ry = std::sqrt(t > 0 ? t : 0) * (R[1] < 0 ? -1. : 1.);
        t = (R[8] + 1) * 0.5; rz = std::sqrt(t > 0 ? t : 0) * (R[2] < 0 ? -1. : 1.);
        if (std::fabs(rx) < std::fabs(ry) && std::fabs(rx) < std::fabs(rz) && (R[5] > 0) != (ry * rz > 0)) rz = -rz;
        theta /= std::sqrt(rx * rx + ry * ry + rz * rz);
        r[0] = rx * theta; r[1] = ry * theta; r[2] = rz * theta;
        return;
    }
    const double vth = 1 / (2 * s) * theta;
    r[0] = rx * vth; r[1] = ry * vth; r[2] = rz * vth;
}

struct RectF { float x, y, width, height; };

// cvUndistortPoints with zero distortion: pixel -> normalised -> R -> new camera matrix (3x3 part of P), float out
static void undistort_point(const double* K, const double* RR, float px, float py, float* ox, float* oy)
{
    const double ifx = 1. / K[0], ify = 1. / K[4];
    double x = (px - K[2]) * ifx, y = (py - K[5]) * ify;
    if (RR) {
        const double xx = RR[0] * x + RR[1] * y + RR[2], yy = RR[3] * x + RR[4] * y + RR[5], ww = 1. / (RR[6] * x + RR[7] * y + RR[8]);
        x = xx * ww; y = yy * ww;
    }
    *ox = (float)x; *oy = (float)y;
}

// icvGetRectangles (calibration.cpp): 9x9 grid over [0,W-1]x[0,H-1]
static void get_rectangles(const double* K, const double* R, const double* P, int W, int H, RectF& inner, RectF& outer)
{
    const int N = 9;
    const double P33[9] = { P[0], P[1], P[2], P[4], P[5], P[6], P[8], P[9], P[10] };
    double RR[9];
    mul33(P33, R, RR);
    float iX0 = -FLT_MAX, iX1 = FLT_MAX, iY0 = -FLT_MAX, iY1 = FLT_MAX;
    float oX0 = FLT_MAX, oX1 = -FLT_MAX, oY0 = FLT_MAX, oY1 = -FLT_MAX;
    for (int y = 0; y < N; ++y)
        for (int x = 0; x < N; ++x) {
            float px, py;
            undistort_point(K, RR, (float)x * (W - 1) / (N - 1), (float)y * (H - 1) / (N - 1), &px, &py);
            oX0 = std::min(oX0, px); oX1 = std::max(oX1, px); oY0 = std::min(oY0, py); oY1 = std::max(oY1, py);
            if (x == 0) iX0 = std::max(iX0, px);
            if (x == N - 1) iX1 = std::min(iX1, px);
            if (y == 0) iY0 = std::max(iY0, py);
            if (y == N - 1) iY1 = std::min(iY1, py);
        }
    inner = RectF{ iX0, iY0, iX1 - iX0, iY1 - iY0 };
    outer = RectF{ oX0, oY0, oX1 - oX0, oY1 - oY0 };
}

static void roi_clip(double x, double y, double w, double h, int W, int H, int* roi)
{
    int rx = (int)std::ceil(x), ry = (int)std::ceil(y), rw = (int)std::floor(w), rh = (int)std::floor(h);
    // cv::Rect & cv::Rect(0,0,W,H)
    const int x1 = std::max(rx, 0), y1 = std::max(ry, 0);
    int w1 = std::min(rx + rw, W) - x1, h1 = std::min(ry + rh, H) - y1;
    if (w1 <= 0 || h1 <= 0) { roi[0] = roi[1] = roi[2] = roi[3] = 0; return; }
    roi[0] = x1; roi[1] = y1; roi[2] = w1; roi[3] = h1;
}

static int launch_resample(wass_ctx* c, bool cubic, const uint8_t* d_src, int sw, int sh, size_t ss, const float* d_mx,
                           const float* d_my, const double* H, int dw, int dh, const int* roi, uint8_t* d_dst)
{
    if (!c) return WASS_ERR_INVALID_ARG;
    if (!d_src || !d_dst || sw <= 0 || sh <= 0 || dw <= 0 || dh <= 0 || ss < (size_t)sw)
        return set_err(c, WASS_ERR_INVALID_ARG, "bad argument");
    if (sw > 32767 || sh > 32767) return set_err(c, WASS_ERR_UNSUPPORTED, "source images larger than 32767 px are not supported");
    int ox = 0, oy = 0, ow = dw, oh = dh;
    if (roi) {
        ox = roi[0]; oy = roi[1]; ow = roi[2]; oh = roi[3];
        if (ox < 0 || oy < 0 || ow <= 0 || oh <= 0 || ox + ow > dw || oy + oh > dh)
            return set_err(c, WASS_ERR_INVALID_ARG, "roi {%d,%d,%d,%d} outside the %dx%d destination", ox, oy, ow, oh, dw, dh);
    }
    int rc = ensure_tables(c);
    if (rc) return rc;
    // a source (or a map) that is still on its way up through wass_upload_async
    if ((rc = wait_uploads(c, d_src, c->stream)) || (d_mx && (rc = wait_uploads(c, d_mx, c->stream))) ||
        (d_my && (rc = wait_uploads(c, d_my, c->stream))))
        return rc;
    const dim3 block(64, 4), grid((ow + 63) / 64, (oh + 3) / 4);
    const short* tab = (const short*)c->rect_tab.p;
    if (cubic) {
        if (!d_mx || !d_my) return set_err(c, WASS_ERR_INVALID_ARG, "null map");
        hipLaunchKernelGGL(k_remap_cubic, grid, block, 0, c->stream, d_src, sw, sh, ss, d_mx, d_my, dw, ox, oy, ow, oh, d_dst, tab);
    } else {
        if (!H) return set_err(c, WASS_ERR_INVALID_ARG, "null homography");
        M9 M;
        if (!inv33(H, M.m)) return set_err(c, WASS_ERR_INVALID_ARG, "singular homography");
        // block width of WarpPerspectiveInvoker: BLOCK_SZ = 32, bh0 = min(16, height), bw0 = min(1024 / bh0, width)
        const int bh0 = std::min(16, dh), bw0 = std::min(32 * 32 / bh0, dw);
        hipLaunchKernelGGL(k_warp_linear, grid, block, 0, c->stream, d_src, sw, sh, ss, M, bw0, ox, oy, ow, oh, d_dst, tab);
    }
    WASS_HIP(c, hipGetLastError());
    return WASS_OK;
}

static int resample_host(wass_ctx* c, bool cubic, const uint8_t* src, int sw, int sh, size_t ss, const float* mx, const float* my,
                         const double* H, int dw, int dh, const int* roi, uint8_t* dst)
{
    if (!c) return WASS_ERR_INVALID_ARG;
    if (!src || !dst || sw <= 0 || sh <= 0 || dw <= 0 || dh <= 0 || ss < (size_t)sw) return set_err(c, WASS_ERR_INVALID_ARG, "bad argument");
    if (roi && (roi[2] <= 0 || roi[3] <= 0)) return set_err(c, WASS_ERR_INVALID_ARG, "empty roi");
    const size_t nsrc = (size_t)sw * sh, nmap = (size_t)dw * dh, nout = roi ? (size_t)roi[2] * roi[3] : nmap;
    int rc;
    if ((rc = ensure(c, c->tmp_in0, nsrc)) || (rc = ensure(c, c->tmp_in1, nout))) return rc;
    WASS_HIP(c, hipMemcpy2DAsync(c->tmp_in0.p, (size_t)sw, src, ss, (size_t)sw, (size_t)sh, hipMemcpyHostToDevice, c->stream));
    if (cubic) {
        if (!mx || !my) return set_err(c, WASS_ERR_INVALID_ARG, "null map");
        if ((rc = ensure(c, c->rect_mx, nmap * 4)) || (rc = ensure(c, c->rect_my, nmap * 4))) return rc;
        WASS_HIP(c, hipMemcpyAsync(c->rect_mx.p, mx, nmap * 4, hipMemcpyHostToDevice, c->stream));
        WASS_HIP(c, hipMemcpyAsync(c->rect_my.p, my, nmap * 4, hipMemcpyHostToDevice, c->stream));
    }
    rc = launch_resample(c, cubic, (const uint8_t*)c->tmp_in0.p, sw, sh, (size_t)sw, (const float*)c->rect_mx.p, (const float*)c->rect_my.p,
                         H, dw, dh, roi, (uint8_t*)c->tmp_in1.p);
    if (rc) return rc;
    WASS_HIP(c, hipMemcpyAsync(dst, c->tmp_in1.p, nout, hipMemcpyDeviceToHost, c->stream));
    WASS_HIP(c, hipStreamSynchronize(c->stream));
    return WASS_OK;
}

static int undistort_dev(wass_ctx* c, const uint8_t* d_src, int w, int h, size_t ss, const double* K, const double* dist, int n,
                         uint8_t* d_dst)
{
    if (!c) return WASS_ERR_INVALID_ARG;
    if (!d_src || !d_dst || !K || (!dist && n) || w <= 0 || h <= 0 || ss < (size_t)w) return set_err(c, WASS_ERR_INVALID_ARG, "bad argument");
    if (!(n == 0 || n == 4 || n == 5 || n == 8 || n == 12))
        return set_err(c, WASS_ERR_UNSUPPORTED, "%d distortion coefficients (4, 5, 8 or 12 supported; no tilt model)", n);
    if (w > 32767 || h > 32767) return set_err(c, WASS_ERR_UNSUPPORTED, "images larger than 32767 px are not supported");
    Dist12 D;
    for (int i = 0; i < 12; ++i) D.k[i] = i < n ? dist[i] : 0.0;
    int rc = ensure_tables(c);
    if (rc) return rc;
    if ((rc = wait_uploads(c, d_src, c->stream))) return rc;
    // The normalised coordinates depend on (K, w, h) only: one table per camera is kept on the device, so the per-frame call
    // of a sequence computes nothing on the host and does not synchronise.
    wass_ctx::UndCache* hit = nullptr;
    for (auto& e : c->und_cache)
        if (e.valid && e.w == w && e.h == h && memcmp(e.K, K, sizeof e.K) == 0) hit = &e;
    if (!hit) {
        // exactly as initUndistortRectifyMap produces them inside cv::undistort's stripes
        std::vector<double> xy((size_t)w + h);
        int stripe0 = std::min(std::max(1, 4096 / std::max(w, 1)), h);
        for (int y0 = 0; y0 < h; y0 += stripe0) {
            const int stripe = std::min(stripe0, h - y0);
            double Ar[9], ir[9];
            memcpy(Ar, K, sizeof Ar);
            Ar[5] = K[5] - y0;
            if (!inv33(Ar, ir)) return set_err(c, WASS_ERR_INVALID_ARG, "singular camera matrix");
            if (ir[1] != 0 || ir[3] != 0 || ir[6] != 0 || ir[7] != 0 || ir[8] != 1)
                return set_err(c, WASS_ERR_UNSUPPORTED, "camera matrix with skew is not supported");
            if (y0 == 0) {
                double _x = ir[2];
                for (int j = 0; j < w; ++j, _x += ir[0]) xy[j] = _x * (1. / 1.);
            }
            for (int i = 0; i < stripe; ++i) xy[(size_t)w + y0 + i] = (i * ir[4] + ir[5]) * (1. / 1.);
        }
        wass_ctx::UndCache& e = c->und_cache[c->und_next];
        c->und_next ^= 1;
        // the entry being replaced may still be read by a kernel in flight
        WASS_HIP(c, hipStreamSynchronize(c->stream));
        e.valid = false;
        if ((rc = ensure(c, e.xy, xy.size() * sizeof(double)))) return rc;
        WASS_HIP(c, hipMemcpyAsync(e.xy.p, xy.data(), xy.size() * sizeof(double), hipMemcpyHostToDevice, c->stream));
        WASS_HIP(c, hipStreamSynchronize(c->stream));          // xy is pageable and about to go out of scope
        memcpy(e.K, K, sizeof e.K); e.w = w; e.h = h; e.valid = true;
        hit = &e;
    }
    const double* dxy = (const double*)hit->xy.p;
    hipLaunchKernelGGL(k_undistort, dim3((w + 63) / 64, (h + 3) / 4), dim3(64, 4), 0, c->stream, d_src, w, h, ss, dxy, dxy + w, D,
                       K[0], K[4], K[2], K[5], d_dst, (const short*)c->rect_tab.p);
    WASS_HIP(c, hipGetLastError());
    return WASS_OK;
}

}  // namespace wass

using namespace wass;

extern "C" {

int wass_undistort_dev(wass_ctx* c, const uint8_t* d_src, int w, int h, size_t ss, const double K[9], const double* dist, int n_dist,
                       uint8_t* d_dst)
{
    return undistort_dev(c, d_src, w, h, ss, K, dist, n_dist, d_dst);
}

int wass_undistort(wass_ctx* c, const uint8_t* src, int w, int h, size_t ss, const double K[9], const double* dist, int n_dist, uint8_t* dst)
{
    if (!c) return WASS_ERR_INVALID_ARG;
    if (!src || !dst || w <= 0 || h <= 0 || ss < (size_t)w) return set_err(c, WASS_ERR_INVALID_ARG, "bad argument");
    const size_t n = (size_t)w * h;
    int rc;
    if ((rc = ensure(c, c->tmp_in0, n)) || (rc = ensure(c, c->tmp_in1, n))) return rc;
    WASS_HIP(c, hipMemcpy2DAsync(c->tmp_in0.p, (size_t)w, src, ss, (size_t)w, (size_t)h, hipMemcpyHostToDevice, c->stream));
    if ((rc = undistort_dev(c, (const uint8_t*)c->tmp_in0.p, w, h, (size_t)w, K, dist, n_dist, (uint8_t*)c->tmp_in1.p))) return rc;
    WASS_HIP(c, hipMemcpyAsync(dst, c->tmp_in1.p, n, hipMemcpyDeviceToHost, c->stream));
    WASS_HIP(c, hipStreamSynchronize(c->stream));
    return WASS_OK;
}

int wass_remap_cubic_dev(wass_ctx* c, const uint8_t* d_src, int sw, int sh, size_t ss, const float* d_mx, const float* d_my, int dw, int dh,
                         const int roi[4], uint8_t* d_dst)
{
    return launch_resample(c, true, d_src, sw, sh, ss, d_mx, d_my, nullptr, dw, dh, roi, d_dst);
}

int wass_remap_cubic(wass_ctx* c, const uint8_t* src, int sw, int sh, size_t ss, const float* mx, const float* my, int dw, int dh,
                     const int roi[4], uint8_t* dst)
{
    return resample_host(c, true, src, sw, sh, ss, mx, my, nullptr, dw, dh, roi, dst);
}

int wass_warp_perspective_dev(wass_ctx* c, const uint8_t* d_src, int sw, int sh, size_t ss, const double H[9], int dw, int dh,
                              const int roi[4], uint8_t* d_dst)
{
    return launch_resample(c, false, d_src, sw, sh, ss, nullptr, nullptr, H, dw, dh, roi, d_dst);
}

int wass_warp_perspective(wass_ctx* c, const uint8_t* src, int sw, int sh, size_t ss, const double H[9], int dw, int dh, const int roi[4],
                          uint8_t* dst)
{
    return resample_host(c, false, src, sw, sh, ss, nullptr, nullptr, H, dw, dh, roi, dst);
}

// cv::initUndistortRectifyMap (undistort.dispatch.cpp) with zero distortion and CV_32FC1 maps: the per-row
// accumulation (_x += ir[0] ...) is kept so that the float casts see the same doubles.
int wass_init_rectify_map(const double K[9], const double R[9], const double P[12], int w, int h, float* map_x, float* map_y)
{
    if (!K || !R || !P || !map_x || !map_y || w <= 0 || h <= 0) return WASS_ERR_INVALID_ARG;
    const double P33[9] = { P[0], P[1], P[2], P[4], P[5], P[6], P[8], P[9], P[10] };
    double PR[9], ir[9];
    mul33(P33, R, PR);
    if (!inv33(PR, ir)) return WASS_ERR_INVALID_ARG;
    const double u0 = K[2], v0 = K[5], fx = K[0], fy = K[4];
    for (int i = 0; i < h; ++i) {
        float* m1 = map_x + (size_t)i * w;
        float* m2 = map_y + (size_t)i * w;
        double _x = i * ir[1] + ir[2], _y = i * ir[4] + ir[5], _w = i * ir[7] + ir[8];
        for (int j = 0; j < w; ++j, _x += ir[0], _y += ir[3], _w += ir[6]) {
            const double iw = 1. / _w, x = _x * iw, y = _y * iw;
            m1[j] = (float)(fx * x + u0);
            m2[j] = (float)(fy * y + v0);
        }
    }
    return WASS_OK;
}

// cvStereoRectify (calibration.cpp), flags = 0, zero distortion, newImageSize = imageSize
int wass_stereo_rectify(const double K1[9], const double K2[9], int W, int H, const double R[9], const double T[3], double alpha,
                        double R1[9], double R2[9], double P1[12], double P2[12], int roi1[4], int roi2[4])
{
    if (!K1 || !K2 || !R || !T || !R1 || !R2 || !P1 || !P2 || W <= 0 || H <= 0) return WASS_ERR_INVALID_ARG;
    double om[3], r_r[9], t[3];
    rodrigues_m2v(R, om);
    for (double& v : om) v *= -0.5;                    // each camera takes half of the relative rotation
    rodrigues_v2m(om, r_r);
    mulv(r_r, T, t);
    const int idx = std::fabs(t[0]) > std::fabs(t[1]) ? 0 : 1;
    const double c = t[idx], nt = std::sqrt(t[0] * t[0] + t[1] * t[1] + t[2] * t[2]);
    if (!(nt > 0.0)) return WASS_ERR_INVALID_ARG;
    double uu[3] = { 0, 0, 0 };
    uu[idx] = c > 0 ? 1 : -1;
    // global rotation that takes the (half-rotated) baseline onto the x (or y) axis
    double ww[3] = { t[1] * uu[2] - t[2] * uu[1], t[2] * uu[0] - t[0] * uu[2], t[0] * uu[1] - t[1] * uu[0] };
    const double nw = std::sqrt(ww[0] * ww[0] + ww[1] * ww[1] + ww[2] * ww[2]);
    if (nw > 0.0) { const double k = std::acos(std::fabs(c) / nt) / nw; for (double& v : ww) v *= k; }
    double wR[9];
    rodrigues_v2m(ww, wR);
    mul33T(wR, r_r, R1);
    mul33(wR, r_r, R2);
    mulv(R2, T, t);

    const double fc0 = (K1[(idx ^ 1) * 4] + K2[(idx ^ 1) * 4]) * 0.5;
    double fc_new = fc0, cc[2][2];
    for (int k = 0; k < 2; ++k) {
        const double* A = k == 0 ? K1 : K2;
        const double* Rk = k == 0 ? R1 : R2;
        double sx = 0, sy = 0;
        for (int i = 0; i < 4; ++i) {
            const int j = i < 2 ? 0 : 1;
            float nx_, ny_;
            undistort_point(A, nullptr, (float)((i % 2) * (W - 1)), (float)(j * (H - 1)), &nx_, &ny_);
            // cvProjectPoints2 with zero translation / distortion, camera matrix diag(fc_new, fc_new, 1)
            const double X = nx_, Y = ny_, Z = 1.0;
            const double x = Rk[0] * X + Rk[1] * Y + Rk[2] * Z, y = Rk[3] * X + Rk[4] * Y + Rk[5] * Z, z = Rk[6] * X + Rk[7] * Y + Rk[8] * Z;
            const double iz = z ? 1. / z : 1;
            sx += (double)(float)(x * iz * fc_new + 0.0);
            sy += (double)(float)(y * iz * fc_new + 0.0);
        }
        cc[k][0] = (W - 1) / 2 - sx / 4;              // sic: integer division of (nx-1)/2
        cc[k][1] = (H - 1) / 2 - sy / 4;
    }
    if (idx == 0) cc[0][1] = cc[1][1] = (cc[0][1] + cc[1][1]) * 0.5;      // horizontal stereo: common cy
    else cc[0][0] = cc[1][0] = (cc[0][0] + cc[1][0]) * 0.5;

    memset(P1, 0, 12 * sizeof(double));
    memset(P2, 0, 12 * sizeof(double));
    P1[0] = P1[5] = fc_new; P1[2] = cc[0][0]; P1[6] = cc[0][1]; P1[10] = 1;
    P2[0] = P2[5] = fc_new; P2[2] = cc[1][0]; P2[6] = cc[1][1]; P2[10] = 1;
    P2[idx * 4 + 3] = t[idx] * fc_new;                // baseline * focal length

    alpha = std::min(alpha, 1.);
    RectF in1, out1, in2, out2;
    get_rectangles(K1, R1, P1, W, H, in1, out1);
    get_rectangles(K2, R2, P2, W, H, in2, out2);

    const double cx1_0 = cc[0][0], cy1_0 = cc[0][1], cx2_0 = cc[1][0], cy2_0 = cc[1][1];
    const double cx1 = W * cx1_0 / W, cy1 = H * cy1_0 / H, cx2 = W * cx2_0 / W, cy2 = H * cy2_0 / H;
    double s = 1.;
    if (alpha >= 0) {
        double s0 = std::max(std::max(std::max(cx1 / (cx1_0 - in1.x), cy1 / (cy1_0 - in1.y)), (W - cx1) / (in1.x + in1.width - cx1_0)),
                             (H - cy1) / (in1.y + in1.height - cy1_0));
        s0 = std::max(std::max(std::max(std::max(cx2 / (cx2_0 - in2.x), cy2 / (cy2_0 - in2.y)), (W - cx2) / (in2.x + in2.width - cx2_0)),
                               (H - cy2) / (in2.y + in2.height - cy2_0)), s0);
        double s1 = std::min(std::min(std::min(cx1 / (cx1_0 - out1.x), cy1 / (cy1_0 - out1.y)), (W - cx1) / (out1.x + out1.width - cx1_0)),
                             (H - cy1) / (out1.y + out1.height - cy1_0));
        s1 = std::min(std::min(std::min(std::min(cx2 / (cx2_0 - out2.x), cy2 / (cy2_0 - out2.y)), (W - cx2) / (out2.x + out2.width - cx2_0)),
                               (H - cy2) / (out2.y + out2.height - cy2_0)), s1);
        s = s0 * (1 - alpha) + s1 * alpha;
    }
    fc_new *= s;
    P1[0] = P1[5] = fc_new; P1[2] = cx1; P1[6] = cy1;
    P2[0] = P2[5] = fc_new; P2[2] = cx2; P2[6] = cy2;
    P2[idx * 4 + 3] = s * P2[idx * 4 + 3];
    if (roi1) roi_clip((in1.x - cx1_0) * s + cx1, (in1.y - cy1_0) * s + cy1, in1.width * s, in1.height * s, W, H, roi1);
    if (roi2) roi_clip((in2.x - cx2_0) * s + cx2, (in2.y - cy2_0) * s + cy2, in2.width * s, in2.height * s, W, H, roi2);
    return WASS_OK;
}

}  // extern "C"
