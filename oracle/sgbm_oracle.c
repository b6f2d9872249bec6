/*
 * sgbm_oracle.c -- CPU restatement of cv::StereoSGBM::compute (OpenCV 4.5.5,
 * MODE_SGBM and MODE_HH) as wass_stereo drives it
 * (reference: src/wass_stereo/wass_stereo.cpp:775-782,837).
 *
 * TEST INFRASTRUCTURE ONLY (see wass_oracle.h).  PARITY UNPINNED: OpenCV is a
 * third-party dependency (pinned 4.5.5 in meta.yaml:12-13,20-21) whose source
 * is not in /root/reference; this follows SURVEY.md Appendix A.1-A.7, i.e. the
 * published algorithm of modules/calib3d/src/stereosgbm.cpp
 * (calcPixelCostBT, computeDisparitySGBM, StereoSGBMImpl::compute) and
 * modules/imgproc median blur (3x3, CV_16S).  Scalar-code semantics: every
 * store to a cost buffer is an (int16) cast (wrap), S uses saturate_cast.
 */
#include "wass_oracle.h"

#include <stdlib.h>
#include <string.h>

#define ORC_MAX_COST 32767
#define ORC_DISP_SHIFT 4
#define ORC_DISP_SCALE 16

typedef int16_t cost_t;

static inline int imin(int a, int b) { return a < b ? a : b; }
static inline int imax(int a, int b) { return a > b ? a : b; }
static inline cost_t sat16(int v) { return (cost_t)(v < -32768 ? -32768 : (v > 32767 ? 32767 : v)); }

/* ------------------------------------------------------------------------
 * A.2  pixel cost of one image row (calcPixelCostBT).
 * chan[0..3] = sobel1, raw1, sobel2, raw2, each w bytes; v0/v1 scratch w bytes.
 * pix[x*D + d] for x in [0,width1), d in [0,D)   (column = x+minX1, disp = d+minD)
 * ------------------------------------------------------------------------ */
static void bt_row(const uint8_t* img1, const uint8_t* img2, int w, int h, int y,
                   int minD, int maxD, const uint8_t* tab0 /* tab[0] at tab0[0] */,
                   uint8_t* chan, uint8_t* v0, uint8_t* v1, cost_t* pix)
{
    const int D = maxD - minD;
    const int minX1 = imax(maxD, 0), maxX1 = w + imin(minD, 0);
    const int width1 = maxX1 - minX1;
    const int minX2 = imax(minX1 - maxD, 0), maxX2 = imin(maxX1 - minD, w);
    uint8_t* c1[2] = { chan, chan + w };
    uint8_t* c2[2] = { chan + 2 * w, chan + 3 * w };
    const uint8_t* r1 = img1 + (size_t)y * w;
    const uint8_t* r2 = img2 + (size_t)y * w;
    const int n = y > 0 ? -w : 0, s = y < h - 1 ? w : 0;   /* row replicate */
    int x, c, d;

    /* both channels of both images: first and last column = tab[0] */
    for (c = 0; c < 2; c++) {
        c1[c][0] = c1[c][w - 1] = tab0[0];
        c2[c][0] = c2[c][w - 1] = tab0[0];
    }
    {
        int lo = imax(imin(minX1, minX2) - 1, 1);
        int hi = imin(imax(maxX1, maxX2) + 1, w - 1);
        for (x = lo; x < hi; x++) {
            c1[0][x] = tab0[(r1[x + 1] - r1[x - 1]) * 2 + r1[x + n + 1] - r1[x + n - 1] + r1[x + s + 1] - r1[x + s - 1]];
            c2[0][x] = tab0[(r2[x + 1] - r2[x - 1]) * 2 + r2[x + n + 1] - r2[x + n - 1] + r2[x + s + 1] - r2[x + s - 1]];
            c1[1][x] = r1[x];
            c2[1][x] = r2[x];
        }
    }
    memset(pix, 0, (size_t)width1 * D * sizeof(cost_t));

    for (c = 0; c < 2; c++) {
        const int diff_scale = c == 0 ? 0 : 2;
        const uint8_t* p1 = c1[c];
        const uint8_t* p2 = c2[c];
        /* half-pixel interval of the second image */
        for (x = minX2; x < maxX2; x++) {
            int v = p2[x];
            int va = x < w - 1 ? (v + p2[x + 1]) / 2 : v;
            int vb = x > 0 ? (v + p2[x - 1]) / 2 : v;
            v0[x] = (uint8_t)imin(imin(va, vb), v);
            v1[x] = (uint8_t)imax(imax(va, vb), v);
        }
        for (x = minX1; x < maxX1; x++) {
            int u = p1[x];
            int ul = x > 0 ? (u + p1[x - 1]) / 2 : u;
            int ur = x < w - 1 ? (u + p1[x + 1]) / 2 : u;
            int u0 = imin(imin(ul, ur), u);
            int u1 = imax(imax(ul, ur), u);
            cost_t* cost = pix + (size_t)(x - minX1) * D;
            for (d = minD; d < maxD; d++) {
                int v = p2[x - d];
                int a0 = imax(imax(0, u - v1[x - d]), v0[x - d] - u);
                int a1 = imax(imax(0, v - u1), u0 - v);
                cost[d - minD] = (cost_t)(cost[d - minD] + (imin(a0, a1) >> diff_scale));
            }
        }
    }
}

/* A.6: cv::medianBlur(3) on int16, replicate border */
void orc_median3_i16(const int16_t* src, int16_t* dst, int w, int h)
{
    int x, y, i, j, k;
    for (y = 0; y < h; y++) {
        const int16_t* rows[3];
        rows[0] = src + (size_t)imax(y - 1, 0) * w;
        rows[1] = src + (size_t)y * w;
        rows[2] = src + (size_t)imin(y + 1, h - 1) * w;
        for (x = 0; x < w; x++) {
            int16_t v[9];
            int xs[3];
            xs[0] = imax(x - 1, 0); xs[1] = x; xs[2] = imin(x + 1, w - 1);
            k = 0;
            for (j = 0; j < 3; j++) for (i = 0; i < 3; i++) v[k++] = rows[j][xs[i]];
            for (i = 1; i < 9; i++) {          /* insertion sort */
                int16_t t = v[i];
                for (j = i - 1; j >= 0 && v[j] > t; j--) v[j + 1] = v[j];
                v[j + 1] = t;
            }
            dst[(size_t)y * w + x] = v[4];
        }
    }
}

int orc_sgbm_compute(const uint8_t* img1, const uint8_t* img2, int w, int h,
                     const orc_sgbm_params* p, int16_t* disp16,
                     int16_t* C_out, int16_t* S_out, int16_t* raw_out,
                     orc_sgbm_stats* stats)
{
    /* A.1 derived parameters */
    const int minD = p->min_disp, maxD = minD + p->num_disp, D = p->num_disp;
    const int uniq = p->uniqueness_ratio >= 0 ? p->uniqueness_ratio : 10;
    const int d12 = p->disp12_max_diff > 0 ? p->disp12_max_diff : 1;
    const int P1 = p->P1 > 0 ? p->P1 : 2;
    const int P2 = imax(p->P2 > 0 ? p->P2 : 5, P1 + 1);
    const int minX1 = imax(maxD, 0), maxX1 = w + imin(minD, 0);
    const int width1 = maxX1 - minX1;
    const int INVALID_SCALED = (minD - 1) * ORC_DISP_SCALE;
    const int win = p->block_size > 0 ? p->block_size : 5;
    const int SW2 = win / 2, SH2 = win / 2;
    const int ftzero = imax(p->prefilter_cap, 15) | 1;
    const int npasses = p->mode == 8 ? 2 : 1;
    const int full = npasses == 2;
    enum { TAB_OFS = 256 * 4, TAB_SIZE = 256 + TAB_OFS * 2 };
    uint8_t tab[TAB_SIZE];
    int k, x, y, d, pass;
    int maxC = 0, maxL = 0;
    int16_t* raw;

    if (stats) { stats->max_C = stats->max_L = stats->overflow = 0; }
    if (w <= 0 || h <= 0 || D <= 0 || (D % 16) != 0 || minD < 0 || (p->mode != 5 && p->mode != 8))
        return -1;

    raw = raw_out ? raw_out : (int16_t*)malloc((size_t)w * h * sizeof(int16_t));
    if (!raw) return -3;

    if (minX1 >= maxX1) {
        for (size_t i = 0; i < (size_t)w * h; i++) raw[i] = (int16_t)INVALID_SCALED;
        orc_median3_i16(raw, disp16, w, h);
        if (!raw_out) free(raw);
        return 0;
    }
    if (width1 <= SW2) { if (!raw_out) free(raw); return -4; }  /* original reads past the row here */

    for (k = 0; k < TAB_SIZE; k++)
        tab[k] = (uint8_t)(imin(imax(k - TAB_OFS, -ftzero), ftzero) + ftzero);

    {
        const size_t rowC = (size_t)width1 * D;
        const int hsumRows = SH2 * 2 + 2;
        const size_t LrW = (size_t)(D + 2);
        const size_t LrRow = (size_t)(width1 + 2) * 4 * LrW;
        const size_t mLRow = (size_t)(width1 + 2) * 4;
        cost_t* Cbuf = (cost_t*)malloc((full ? (size_t)h : 1) * rowC * sizeof(cost_t));
        cost_t* Sbuf = (cost_t*)malloc((full ? (size_t)h : 1) * rowC * sizeof(cost_t));
        cost_t* hsum = (cost_t*)malloc((size_t)hsumRows * rowC * sizeof(cost_t));
        cost_t* pix = (cost_t*)malloc(rowC * sizeof(cost_t));
        cost_t* Lr = (cost_t*)malloc(2 * LrRow * sizeof(cost_t));
        cost_t* mLr = (cost_t*)malloc(2 * mLRow * sizeof(cost_t));
        uint8_t* chan = (uint8_t*)malloc((size_t)w * 6);
        int16_t* disp2 = (int16_t*)malloc((size_t)w * sizeof(int16_t));
        cost_t* disp2cost = (cost_t*)malloc((size_t)w * sizeof(cost_t));
        if (!Cbuf || !Sbuf || !hsum || !pix || !Lr || !mLr || !chan || !disp2 || !disp2cost) return -3;

#define CROW(yy) (Cbuf + (full ? (size_t)(yy) * rowC : 0))
#define SROW(yy) (Sbuf + (full ? (size_t)(yy) * rowC : 0))
#define HSUM(yy) (hsum + (size_t)((yy) % hsumRows) * rowC)
#define LR(id, xx, dir) (Lr + (size_t)(id) * LrRow + ((size_t)((xx) + 1) * 4 + (dir)) * LrW + 1)
#define MLR(id, xx, dir) (mLr + (size_t)(id) * mLRow + (size_t)((xx) + 1) * 4 + (dir))

        /* initCBuf(P2): "add P2 to every C(x,y)" */
        for (size_t i = 0; i < (full ? (size_t)h : 1) * rowC; i++) Cbuf[i] = (cost_t)P2;

        for (pass = 1; pass <= npasses; pass++) {
            int x1, x2, y1, y2, dx, dy, lrID = 0;
            if (pass == 1) { y1 = 0; y2 = h; dy = 1; x1 = 0; x2 = width1; dx = 1; }
            else { y1 = h - 1; y2 = -1; dy = -1; x1 = width1 - 1; x2 = -1; dx = -1; }
            memset(Lr, 0, 2 * LrRow * sizeof(cost_t));      /* clearLr */
            memset(mLr, 0, 2 * mLRow * sizeof(cost_t));

            for (y = y1; y != y2; y += dy) {
                int16_t* disp1 = raw + (size_t)y * w;
                cost_t* C = CROW(y);
                cost_t* S = SROW(y);

                if (pass == 1) {
                    /* A.3: block sum, sliding in y */
                    int dy1 = y == 0 ? 0 : y + SH2, dy2 = y == 0 ? SH2 : dy1;
                    for (k = dy1; k <= dy2; k++) {
                        cost_t* hAdd = HSUM(imin(k, h - 1));
                        if (k < h) {
                            bt_row(img1, img2, w, h, k, minD, maxD, tab + TAB_OFS, chan, chan + 4 * w, chan + 5 * w, pix);
                            /* horizontal sliding window, replicate in the width1 domain */
                            memset(hAdd, 0, (size_t)D * sizeof(cost_t));
                            for (x = 0; x <= SW2; x++) {
                                int scale = x == 0 ? SW2 + 1 : 1;
                                for (d = 0; d < D; d++)
                                    hAdd[d] = (cost_t)(hAdd[d] + pix[(size_t)x * D + d] * scale);
                            }
                            if (y > 0) {
                                const cost_t* hSub = HSUM(imax(y - SH2 - 1, 0));
                                const cost_t* Cprev = CROW(y - 1);
                                for (d = 0; d < D; d++) {
                                    int v = Cprev[d] + hAdd[d] - hSub[d];
                                    if (v > maxC) maxC = v;
                                    C[d] = (cost_t)v;
                                }
                                for (x = 1; x < width1; x++) {
                                    const cost_t* pa = pix + (size_t)imin(x + SW2, width1 - 1) * D;
                                    const cost_t* ps = pix + (size_t)imax(x - SW2 - 1, 0) * D;
                                    cost_t* hx = hAdd + (size_t)x * D;
                                    for (d = 0; d < D; d++) {
                                        int hv = hx[d] = (cost_t)(hx[d - D] + pa[d] - ps[d]);
                                        int v = Cprev[(size_t)x * D + d] + hv - hSub[(size_t)x * D + d];
                                        if (v > maxC) maxC = v;
                                        C[(size_t)x * D + d] = (cost_t)v;
                                    }
                                }
                            } else {
                                int scale = k == 0 ? SH2 + 1 : 1;
                                for (d = 0; d < D; d++) {
                                    int v = C[d] + hAdd[d] * scale;
                                    if (v > maxC) maxC = v;
                                    C[d] = (cost_t)v;
                                }
                                for (x = 1; x < width1; x++) {
                                    const cost_t* pa = pix + (size_t)imin(x + SW2, width1 - 1) * D;
                                    const cost_t* ps = pix + (size_t)imax(x - SW2 - 1, 0) * D;
                                    cost_t* hx = hAdd + (size_t)x * D;
                                    for (d = 0; d < D; d++) {
                                        int hv = hx[d] = (cost_t)(hx[d - D] + pa[d] - ps[d]);
                                        int v = C[(size_t)x * D + d] + hv * scale;
                                        if (v > maxC) maxC = v;
                                        C[(size_t)x * D + d] = (cost_t)v;
                                    }
                                }
                            }
                        } else {
                            /* below the image: re-add the last row's hsum (row replicate) */
                            if (y > 0) {
                                const cost_t* hSub = HSUM(imax(y - SH2 - 1, 0));
                                const cost_t* Cprev = CROW(y - 1);
                                for (size_t i = 0; i < rowC; i++) {
                                    int v = Cprev[i] + hAdd[i] - hSub[i];
                                    if (v > maxC) maxC = v;
                                    C[i] = (cost_t)v;
                                }
                            } else {
                                for (size_t i = 0; i < rowC; i++) {
                                    int v = C[i] + hAdd[i];
                                    if (v > maxC) maxC = v;
                                    C[i] = (cost_t)v;
                                }
                            }
                        }
                    }
                    memset(S, 0, rowC * sizeof(cost_t));     /* clearSBuf */
                    if (C_out)
                        for (size_t i = 0; i < rowC; i++) C_out[(size_t)y * rowC + i] = (cost_t)(C[i] - P2);
                }

                /* A.4: four paths of this pass */
                for (x = x1; x != x2; x += dx) {
                    const int delta0 = P2 + *MLR(lrID, x - dx, 0);
                    const int delta1 = P2 + *MLR(1 - lrID, x - 1, 1);
                    const int delta2 = P2 + *MLR(1 - lrID, x, 2);
                    const int delta3 = P2 + *MLR(1 - lrID, x + 1, 3);
                    cost_t* Lp0 = LR(lrID, x - dx, 0);
                    cost_t* Lp1 = LR(1 - lrID, x - 1, 1);
                    cost_t* Lp2 = LR(1 - lrID, x, 2);
                    cost_t* Lp3 = LR(1 - lrID, x + 1, 3);
                    cost_t* L0o = LR(lrID, x, 0);
                    cost_t* L1o = LR(lrID, x, 1);
                    cost_t* L2o = LR(lrID, x, 2);
                    cost_t* L3o = LR(lrID, x, 3);
                    const cost_t* Cp = C + (size_t)x * D;
                    cost_t* Sp = S + (size_t)x * D;
                    int m0 = ORC_MAX_COST, m1 = ORC_MAX_COST, m2 = ORC_MAX_COST, m3 = ORC_MAX_COST;
                    Lp0[-1] = Lp0[D] = ORC_MAX_COST;
                    Lp1[-1] = Lp1[D] = ORC_MAX_COST;
                    Lp2[-1] = Lp2[D] = ORC_MAX_COST;
                    Lp3[-1] = Lp3[D] = ORC_MAX_COST;
                    for (d = 0; d < D; d++) {
                        int Cpd = Cp[d], Spd = Sp[d], L;
                        L = Cpd + imin((int)Lp0[d], imin(Lp0[d - 1] + P1, imin(Lp0[d + 1] + P1, delta0))) - delta0;
                        if (L > maxL) maxL = L;
                        L0o[d] = (cost_t)L; m0 = imin(m0, L); Spd += L;
                        L = Cpd + imin((int)Lp1[d], imin(Lp1[d - 1] + P1, imin(Lp1[d + 1] + P1, delta1))) - delta1;
                        if (L > maxL) maxL = L;
                        L1o[d] = (cost_t)L; m1 = imin(m1, L); Spd += L;
                        L = Cpd + imin((int)Lp2[d], imin(Lp2[d - 1] + P1, imin(Lp2[d + 1] + P1, delta2))) - delta2;
                        if (L > maxL) maxL = L;
                        L2o[d] = (cost_t)L; m2 = imin(m2, L); Spd += L;
                        L = Cpd + imin((int)Lp3[d], imin(Lp3[d - 1] + P1, imin(Lp3[d + 1] + P1, delta3))) - delta3;
                        if (L > maxL) maxL = L;
                        L3o[d] = (cost_t)L; m3 = imin(m3, L); Spd += L;
                        Sp[d] = sat16(Spd);
                    }
                    *MLR(lrID, x, 0) = (cost_t)m0;
                    *MLR(lrID, x, 1) = (cost_t)m1;
                    *MLR(lrID, x, 2) = (cost_t)m2;
                    *MLR(lrID, x, 3) = (cost_t)m3;
                }

                if (pass == npasses) {
                    /* A.5: disparity selection for this row */
                    for (x = 0; x < w; x++) {
                        disp1[x] = disp2[x] = (int16_t)INVALID_SCALED;
                        disp2cost[x] = ORC_MAX_COST;
                    }
                    for (x = width1 - 1; x >= 0; x--) {
                        cost_t* Sp = S + (size_t)x * D;
                        int minS = ORC_MAX_COST, bestDisp = -1;
                        if (npasses == 1) {
                            /* fifth path (right to left) fused with the search */
                            cost_t* Lp0 = LR(lrID, x + 1, 0);
                            cost_t* L0o = LR(lrID, x, 0);
                            const cost_t* Cp = C + (size_t)x * D;
                            const int delta0 = P2 + *MLR(lrID, x + 1, 0);
                            int m0 = ORC_MAX_COST;
                            Lp0[-1] = Lp0[D] = ORC_MAX_COST;
                            for (d = 0; d < D; d++) {
                                int L0 = Cp[d] + imin((int)Lp0[d], imin(Lp0[d - 1] + P1, imin(Lp0[d + 1] + P1, delta0))) - delta0;
                                int Sval;
                                if (L0 > maxL) maxL = L0;
                                L0o[d] = (cost_t)L0;
                                m0 = imin(m0, L0);
                                Sval = Sp[d] = sat16(Sp[d] + L0);
                                if (Sval < minS) { minS = Sval; bestDisp = d; }
                            }
                            *MLR(lrID, x, 0) = (cost_t)m0;
                        } else {
                            for (d = 0; d < D; d++) {
                                int Sval = Sp[d];
                                if (Sval < minS) { minS = Sval; bestDisp = d; }
                            }
                        }
                        for (d = 0; d < D; d++)
                            if (Sp[d] * (100 - uniq) < minS * 100 && abs(bestDisp - d) > 1) break;
                        if (d < D) continue;
                        d = bestDisp;
                        {
                            int X2 = x + minX1 - d - minD;
                            if (disp2cost[X2] > minS) {
                                disp2cost[X2] = (cost_t)minS;
                                disp2[X2] = (int16_t)(d + minD);
                            }
                        }
                        if (0 < d && d < D - 1) {
                            int denom2 = imax(Sp[d - 1] + Sp[d + 1] - 2 * Sp[d], 1);
                            d = d * ORC_DISP_SCALE + ((Sp[d - 1] - Sp[d + 1]) * ORC_DISP_SCALE + denom2) / (denom2 * 2);
                        } else
                            d *= ORC_DISP_SCALE;
                        disp1[x + minX1] = (int16_t)(d + minD * ORC_DISP_SCALE);
                    }
                    for (x = minX1; x < maxX1; x++) {
                        int d1 = disp1[x], dlo, dhi, xlo, xhi;
                        if (d1 == INVALID_SCALED) continue;
                        dlo = d1 >> ORC_DISP_SHIFT;
                        dhi = (d1 + ORC_DISP_SCALE - 1) >> ORC_DISP_SHIFT;
                        xlo = x - dlo; xhi = x - dhi;
                        if (0 <= xlo && xlo < w && disp2[xlo] >= minD && abs(disp2[xlo] - dlo) > d12 &&
                            0 <= xhi && xhi < w && disp2[xhi] >= minD && abs(disp2[xhi] - dhi) > d12)
                            disp1[x] = (int16_t)INVALID_SCALED;
                    }
                    if (S_out) memcpy(S_out + (size_t)y * rowC, S, rowC * sizeof(cost_t));
                }
                lrID = 1 - lrID;
            }
        }
        free(Cbuf); free(Sbuf); free(hsum); free(pix); free(Lr); free(mLr);
        free(chan); free(disp2); free(disp2cost);
    }

    /* A.6 */
    orc_median3_i16(raw, disp16, w, h);
    /* StereoSGBMImpl::compute: filterSpeckles(disp, (minDisparity - 1) * 16, speckleWindowSize, 16 * speckleRange) */
    if (p->speckle_window > 0) orc_filter_speckles(disp16, w, h, INVALID_SCALED, p->speckle_window, 16 * p->speckle_range);
    if (!raw_out) free(raw);
    if (stats) {
        stats->max_C = maxC; stats->max_L = maxL;
        stats->overflow = (maxC > 32767 || maxL > 32767);
    }
    return 0;
}
