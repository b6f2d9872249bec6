/*
 * clahe_oracle.c -- CPU restatement of cv::CLAHE::apply for CV_8UC1, as wass_prepare calls it
 * (src/wass_prepare/wass_prepare.cpp:257-262,446-449: createCLAHE(CAMx_CLAHE_CLIPLIMIT, Size(T, T)), T = CAMx_CLAHE_TILEGRIDSIZE).
 *
 * TEST INFRASTRUCTURE ONLY (see wass_oracle.h).  PARITY UNPINNED: OpenCV 4.5.5 modules/imgproc/src/clahe.cpp is not in
 * /root/reference and not installed; restated from the published algorithm:
 *   - the image is extended to a multiple of the tile grid with BORDER_REFLECT_101 (bottom / right) when it is not one;
 *   - per tile: 256-bin histogram, clipped at max(1, (int)(clipLimit * tileArea / 256)), the excess redistributed
 *     (excess / 256 to every bin, the remainder one count each to bins 0, step, 2 step, ... with step = max(256 / remainder, 1));
 *     LUT[i] = saturate_cast<uchar>(cumulative(i) * 255 / tileArea)  (float product, round half to even);
 *   - per pixel of the ORIGINAL image: bilinear blend of the four neighbouring tiles' LUT values, tile coordinates
 *     x / tileWidth - 0.5 (float), indices clamped to the grid, result rounded half to even.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "wass_oracle.h"

static int refl101c(int i, int n) { while (i < 0 || i >= n) { if (i < 0) i = -i; else i = 2 * n - 2 - i; if (n == 1) return 0; } return i; }

int orc_clahe(const uint8_t* src, int w, int h, size_t stride, double clip_limit, int tiles_x, int tiles_y, uint8_t* dst)
{
    int ew, eh, tw, th, area, clip = 0, tx, ty, x, y, i;
    uint8_t* lut;
    float lut_scale, inv_tw, inv_th;
    if (w <= 0 || h <= 0 || tiles_x <= 0 || tiles_y <= 0) return -1;
    if (w % tiles_x == 0 && h % tiles_y == 0) { ew = w; eh = h; }
    else { ew = w + (tiles_x - w % tiles_x); eh = h + (tiles_y - h % tiles_y); }
    tw = ew / tiles_x; th = eh / tiles_y;
    area = tw * th;
    lut_scale = (float)255 / (float)area;
    if (clip_limit > 0.0) { clip = (int)(clip_limit * area / 256); if (clip < 1) clip = 1; }
    lut = (uint8_t*)malloc((size_t)tiles_x * tiles_y * 256);
    for (ty = 0; ty < tiles_y; ++ty)
        for (tx = 0; tx < tiles_x; ++tx) {
            int hist[256], sum = 0;
            memset(hist, 0, sizeof hist);
            for (y = 0; y < th; ++y)
                for (x = 0; x < tw; ++x) {
                    const int gx = tx * tw + x, gy = ty * th + y;
                    hist[src[(size_t)refl101c(gy, h) * stride + refl101c(gx, w)]]++;     /* the extension only adds rows / columns past the end */
                }
            if (clip > 0) {
                int clipped = 0, batch, residual;
                for (i = 0; i < 256; ++i) if (hist[i] > clip) { clipped += hist[i] - clip; hist[i] = clip; }
                batch = clipped / 256; residual = clipped - batch * 256;
                for (i = 0; i < 256; ++i) hist[i] += batch;
                if (residual != 0) {
                    const int step = 256 / residual > 1 ? 256 / residual : 1;
                    for (i = 0; i < 256 && residual > 0; i += step, residual--) hist[i]++;
                }
            }
            for (i = 0; i < 256; ++i) {
                long r;
                sum += hist[i];
                r = lrintf((float)sum * lut_scale);
                lut[((size_t)ty * tiles_x + tx) * 256 + i] = (uint8_t)(r < 0 ? 0 : (r > 255 ? 255 : r));
            }
        }
    inv_tw = 1.0f / (float)tw; inv_th = 1.0f / (float)th;
    for (y = 0; y < h; ++y) {
        const float tyf = (float)y * inv_th - 0.5f;
        int ty1 = (int)floorf(tyf), ty2 = ty1 + 1;
        const float ya = tyf - (float)ty1, ya1 = 1.0f - ya;
        if (ty1 < 0) ty1 = 0;
        if (ty2 > tiles_y - 1) ty2 = tiles_y - 1;
        for (x = 0; x < w; ++x) {
            const float txf = (float)x * inv_tw - 0.5f;
            int tx1 = (int)floorf(txf), tx2 = tx1 + 1;
            const float xa = txf - (float)tx1, xa1 = 1.0f - xa;
            const int v = src[(size_t)y * stride + x];
            float res;
            long r;
            if (tx1 < 0) tx1 = 0;
            if (tx2 > tiles_x - 1) tx2 = tiles_x - 1;
            res = ((float)lut[((size_t)ty1 * tiles_x + tx1) * 256 + v] * xa1 + (float)lut[((size_t)ty1 * tiles_x + tx2) * 256 + v] * xa) * ya1 +
                  ((float)lut[((size_t)ty2 * tiles_x + tx1) * 256 + v] * xa1 + (float)lut[((size_t)ty2 * tiles_x + tx2) * 256 + v] * xa) * ya;
            r = lrintf(res);
            dst[(size_t)y * w + x] = (uint8_t)(r < 0 ? 0 : (r > 255 ? 255 : r));
        }
    }
    free(lut);
    return 0;
}
