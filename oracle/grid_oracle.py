"""grid_oracle.py -- CPU restatement of the first gridding step.  TEST INFRASTRUCTURE ONLY.

Follows gridding/wassgridsurface/wassgridsurface.py:316-365 (_grid_task, algorithm "IDW"), wass_utils.py:38-61
(compute_sea_plane_RT / align_on_sea_plane_RT) and IDWInterpolator.py:23-58; the inverse-distance step calls
scipy.signal.convolve2d exactly as the reference does.  cv.morphologyEx(MORPH_CLOSE, 5x5 ones) is OpenCV (absent here):
restated as a 5x5 dilation followed by a 5x5 erosion that ignore what lies outside the image (OpenCV's default border
values for the two operations).

Cell statistic.  The reference fills a cell with the nanmedian of ten random sub-samples scattered last-writer-wins
(:330-345): the result depends on numpy's global random state and on the order of the points.  Two modes:
  cell="mean"              the mean of the cell's points in plain float64 (np.add.at) -- what grid.hip computes up to its
                           fixed-point accumulation; nothing here mirrors the GPU's arithmetic, the test tolerance follows
                           from the GPU's 2^-24 quantisation and its float32 output;
  cell="median"            the exact median of the cell's points (np.median per cell), the deterministic statistic grid.hip offers
                           beside the mean (wass_mesh_grid_idw_ex);
  cell="subsample_median"  the reference's statistic restated line by line (np.random.seed(seed) first), so that the GPU
                           grid can also be compared with what the reference would have produced for one seed."""
import numpy as np
import scipy.signal


def compute_sea_plane_RT(plane):
    a, b, c, d = plane
    q = (1 - c) / (a * a + b * b)
    R = np.array([[1 - a * a * q, -a * b * q, -a], [-a * b * q, 1 - b * b * q, -b], [a, b, c]])
    T = np.array([[0.0], [0.0], [d]])
    return R, T


def cell_values(px, py, pz, width, height, cell="mean", seed=0, subsample_percent=100):
    """(height, width) float64 map of the binned points, NaN where a cell is empty."""
    if cell == "mean":
        cnt = np.zeros((height, width), np.float64)
        ssum = np.zeros((height, width), np.float64)
        np.add.at(cnt, (py, px), 1.0)
        np.add.at(ssum, (py, px), pz)
        with np.errstate(invalid="ignore", divide="ignore"):
            return np.where(cnt > 0, ssum / cnt, np.nan)
    if cell == "median":                                 # exact per-cell median (numpy's: mean of the two middle values for even counts)
        out = np.full((height, width), np.nan)
        key = py * width + px
        order = np.argsort(key, kind="stable")
        ks, zs = key[order], pz[order]
        starts = np.flatnonzero(np.r_[True, ks[1:] != ks[:-1]])
        ends = np.r_[starts[1:], ks.size]
        for a, b in zip(starts, ends):
            out.flat[ks[a]] = np.median(zs[a:b])
        return out
    if cell == "subsample_median":                       # wassgridsurface.py:319,330-345, verbatim order of the random draws
        np.random.seed(seed)
        perm = np.random.permutation(px.shape[0])        # :319 (the permutation of the aligned mesh, here of its in-grid points)
        px, py, pz = px[perm], py[perm], pz[perm]
        NREPS = 10
        ZZ = np.ones([height, width, NREPS], dtype=np.float32) * np.nan
        indices = np.arange(px.shape[0])
        n_pts = int(px.shape[0] * subsample_percent // 100)
        for ii in range(NREPS):
            np.random.shuffle(indices)
            curr_indices = np.copy(indices[:n_pts])
            ZZ[py[indices[curr_indices]], px[indices[curr_indices]], ii] = pz[indices[curr_indices]]
        with np.errstate(all="ignore"):
            import warnings
            with warnings.catch_warnings():
                warnings.simplefilter("ignore", RuntimeWarning)
                return np.nanmedian(ZZ, axis=-1).astype(np.float64)
    raise ValueError(cell)


def grid_idw(points, plane, baseline, xmin, xmax, ymin, ymax, width, height, cell="mean", seed=0, subsample_percent=100):
    """points: (3, N) camera-frame cloud.  Returns (Zi float64 with NaN outside the mask, mask uint8)."""
    R, T = compute_sea_plane_RT(plane)
    m = R @ points + T
    m[2, :] *= -1.0
    m = m * baseline
    px = np.floor((m[0] - xmin) / (xmax - xmin) * (width - 1) + 0.5)
    py = np.floor((m[1] - ymin) / (ymax - ymin) * (height - 1) + 0.5)
    good = (px >= 0) & (px < width) & (py >= 0) & (py < height)
    px, py, pz = px[good].astype(np.int64), py[good].astype(np.int64), m[2, good]
    ZZ = cell_values(px, py, pz, width, height, cell, seed, subsample_percent)
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
