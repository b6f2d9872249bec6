// rectify.hpp -- the reference's built-in rectification (USE_CUSTOM_STEREORECTIFY=true), host side.
//
//   stereoRectifyUndistorted   src/wass_stereo/stereorectify.cpp:57-244
//
// Row f1 of SURVEY.md section 8.  The image resampling (cv::warpPerspective, wass_stereo.cpp:515-516) and the
// whole cv::stereoRectify path (USE_CUSTOM_STEREORECTIFY=false) are entry points of libwassgpu (csrc/rectify.hip).
#pragma once

#include <algorithm>
#include <cmath>

#include "hostio.hpp"

namespace wasshost {

struct Rect { int x = 0, y = 0, width = 0, height = 0; };

inline void cross3(const double* a, const double* b, double* o) { o[0] = a[1] * b[2] - a[2] * b[1]; o[1] = a[2] * b[0] - a[0] * b[2]; o[2] = a[0] * b[1] - a[1] * b[0]; }
inline double norm3(const double* a) { return std::sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]); }

struct HFunctional {          // stereorectify.cpp:70-137
    Mat K0i, K1i, Ri, Rplane, H0, H1;
    HFunctional(const double* ep1, const Mat& K0, const Mat& K1, const Mat& R)
    {
        K0i = inv3(K0); K1i = inv3(K1); Ri = R;
        double Rv[3], N[3], Rk[3];
        const double n = norm3(ep1);
        for (int i = 0; i < 3; ++i) Rv[i] = ep1[i] / n;
        const double ey[3] = { 0, 1, 0 };
        cross3(Rv, ey, N);
        const double nn = norm3(N);
        for (int i = 0; i < 3; ++i) N[i] /= nn;
        cross3(Rv, N, Rk);
        Rplane = Mat(3, 3);
        for (int j = 0; j < 3; ++j) { Rplane(0, j) = Rv[j]; Rplane(1, j) = Rk[j]; Rplane(2, j) = N[j]; }
    }
    double calc(double x)
    {
        const double th = x / 180 * 3.14;                       // sic: 3.14, not pi (:94-95)
        Mat Radd = Mat::eye(3);                                 // Rodrigues((th,0,0)) = rotation about x
        Radd(1, 1) = std::cos(th); Radd(1, 2) = -std::sin(th); Radd(2, 1) = std::sin(th); Radd(2, 2) = std::cos(th);
        H0 = matmul(matmul(Radd, Rplane), K0i);
        H1 = matmul(matmul(matmul(Radd, Rplane), Ri), K1i);
        H0 = scaled(H0, 1.0 / H0(2, 2));
        H1 = scaled(H1, 1.0 / H1(2, 2));
        const double v1 = H0(2, 0) * H0(2, 0) + H0(2, 1) * H0(2, 1);
        const double v2 = H1(2, 0) * H1(2, 0) + H1(2, 1) * H1(2, 1);
        H0 = scaled(H0, 1.0 / std::cbrt(det3(H0)));
        H1 = scaled(H1, 1.0 / std::cbrt(det3(H1)));
        return std::max(v1, v2);
    }
};

inline void stereoRectifyUndistorted(const Mat& K0, const Mat& K1, const Mat& R, const double T[3], double rot_angle, int W, int H,
                                     Mat& H0, Mat& H1, Rect& ROI)
{
    HFunctional hf(T, K0, K1, R);
    double best = rot_angle;
    if (rot_angle == 0) {
        // The reference minimises with cv::DownhillSolver (third party).  Same objective, plain 1-D search:
        // coarse scan then golden-section refinement.  Documented divergence (optimiser, not objective).
        double bx = 0, bv = hf.calc(0);
        for (double a = -90; a <= 90; a += 0.25) { const double v = hf.calc(a); if (v < bv) { bv = v; bx = a; } }
        double lo = bx - 0.25, hi = bx + 0.25;
        const double g = 0.6180339887498949;
        for (int it = 0; it < 80; ++it) {
            const double c = hi - g * (hi - lo), d = lo + g * (hi - lo);
            if (hf.calc(c) < hf.calc(d)) hi = d; else lo = c;
        }
        best = 0.5 * (lo + hi);
        WLOGI << "Best rectifying-plane angle: " << best << " deg.";
    }
    hf.calc(best);
    H0 = hf.H0; H1 = hf.H1;
    auto corners = [&](const Mat& Hm, double px[4], double py[4]) {
        const double cx[4] = { 0, (double)W, (double)W, 0 }, cy[4] = { 0, 0, (double)H, (double)H };
        for (int i = 0; i < 4; ++i) {
            const double x = Hm(0, 0) * cx[i] + Hm(0, 1) * cy[i] + Hm(0, 2), y = Hm(1, 0) * cx[i] + Hm(1, 1) * cy[i] + Hm(1, 2),
                         w = Hm(2, 0) * cx[i] + Hm(2, 1) * cy[i] + Hm(2, 2);
            px[i] = x / w; py[i] = y / w;
        }
    };
    double x0[4], y0[4], x1[4], y1[4];
    corners(H0, x0, y0); corners(H1, x1, y1);
    const double r0x = std::min(x0[0], x0[3]), r0y = std::min(y0[0], y0[1]), r0w = std::max(x0[1], x0[2]) - r0x, r0h = std::max(y0[2], y0[3]) - r0y;
    const double r1x = std::min(x1[0], x1[3]), r1y = std::min(y1[0], y1[1]), r1w = std::max(x1[1], x1[2]) - r1x, r1h = std::max(y1[2], y1[3]) - r1y;
    const double top = std::min(r0y, r1y), bottom = std::max(r0y + r0h, r1y + r1h);
    auto adjust = [&](Mat& Hm, double rx, double rw) {
        Mat Tr = Mat::eye(3); Tr(0, 2) = -rx; Tr(1, 2) = -top;
        Mat Sc = Mat::eye(3); Sc(0, 0) = W / rw; Sc(1, 1) = H / (bottom - top);
        Hm = matmul(matmul(Sc, Tr), Hm);
        Hm = scaled(Hm, 1.0 / std::cbrt(det3(Hm)));
    };
    adjust(H0, r0x, r0w); adjust(H1, r1x, r1w);
    corners(H0, x0, y0); corners(H1, x1, y1);
    double xv[8], yv[8];
    for (int i = 0; i < 4; ++i) { xv[2 * i] = x0[i]; yv[2 * i] = y0[i]; xv[2 * i + 1] = x1[i]; yv[2 * i + 1] = y1[i]; }
    std::sort(xv, xv + 8); std::sort(yv, yv + 8);
    ROI.x = (int)xv[3]; ROI.y = (int)yv[3];                      // implicit double -> int truncation (:240-243)
    ROI.width = (int)(xv[4] - ROI.x); ROI.height = (int)(yv[4] - ROI.y);
}

}  // namespace wasshost
