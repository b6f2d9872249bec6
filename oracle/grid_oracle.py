"""grid_oracle.py -- CPU restatement of the first gridding step.  TEST INFRASTRUCTURE ONLY.

Follows gridding/wassgridsurface/wassgridsurface.py:316-365 (_grid_task, algorithm "IDW"), wass_utils.py:38-61
(compute_sea_plane_RT / align_on_sea_plane_RT) and IDWInterpolator.py:23-58; the inverse-distance step calls
scipy.signal.convolve2d exactly as the reference does.  cv.morphologyEx(MORPH_CLOSE, 5x5 ones) is OpenCV (absent here):
restated as a 5x5 dilation followed by a 5x5 erosion that ignore what lies outside the image (OpenCV's default border
values for the two operations).  The reference's randomised cell value (nanmedian of ten random sub-samples with a
last-writer-wins scatter, :330-345) is replaced by the mean of the cell's points, as in grid.hip."""
import numpy as np
import scipy.signal


def compute_sea_plane_RT(plane):
    a, b, c, d = plane
    q = (1 - c) / (a * a + b * b)
    R = np.array([[1 - a * a * q, -a * b * q, -a], [-a * b * q, 1 - b * b * q, -b], [a, b, c]])
    T = np.array([[0.0], [0.0], [d]])
    return R, T


def grid_idw(points, plane, baseline, xmin, xmax, ymin, ymax, width, height):
    """points: (3, N) camera-frame cloud.  Returns (Zi float64 with NaN outside the mask, mask uint8)."""
    R, T = compute_sea_plane_RT(plane)
    m = R @ points + T
    m[2, :] *= -1.0
    m = m * baseline
    px = np.floor((m[0] - xmin) / (xmax - xmin) * (width - 1) + 0.5)
    py = np.floor((m[1] - ymin) / (ymax - ymin) * (height - 1) + 0.5)
    good = (px >= 0) & (px < width) & (py >= 0) & (py < height)
    px, py, pz = px[good].astype(np.int64), py[good].astype(np.int64), m[2, good]
    cnt = np.zeros((height, width), np.int64)
    ssum = np.zeros((height, width), np.int64)
    np.add.at(cnt, (py, px), 1)
    np.add.at(ssum, (py, px), np.rint(pz * 16777216.0).astype(np.int64))       # the GPU's 2^-24 fixed point
    with np.errstate(invalid="ignore", divide="ignore"):
        ZZ = np.where(cnt > 0, ssum / 16777216.0 / cnt, np.nan)
    # IDWInterpolator(KSIZE=5, exp=2.4, reps=1)
    KS = 5
    Kd = np.array([(k - KS // 2) for k in range(KS)], dtype=np.float64)
    Kx = np.tile(Kd, (KS, 1)); Ky = Kx.T
    with np.errstate(divide="ignore"):
        K = 1.0 / np.power(np.sqrt(Kx ** 2 + Ky ** 2), 2.4)
    K[KS // 2, KS // 2] = 0
    orig = (~np.isnan(ZZ)).astype(np.uint8)
    I = np.where(np.isnan(ZZ), 0.0, ZZ)
    I2 = scipy.signal.convolve2d(I, K, mode="same") / (scipy.signal.convolve2d(orig.astype(np.float32), K, mode="same") + 1e-9)
    out = orig * I + (1 - orig) * I2
    pad = np.pad(orig, 2, constant_values=0)
    dil = np.max([pad[i:i + height, j:j + width] for i in range(5) for j in range(5)], axis=0)
    pad = np.pad(dil, 2, constant_values=1)
    clo = np.min([pad[i:i + height, j:j + width] for i in range(5) for j in range(5)], axis=0)
    out = np.where(clo == 0, np.nan, out)
    return out, clo.astype(np.uint8)
