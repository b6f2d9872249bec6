"""Row f3: surface gridding straight from the device-resident mesh (grid.hip) against oracle/grid_oracle.py."""
import numpy as np
import pytest

import wass_amd
from wass_amd import default_sgm_params, synth

pytestmark = pytest.mark.gpu


def test_grid_matches_oracle_on_a_random_cloud(gpu_ctx):
    from oracle import grid_oracle as G
    rng = np.random.default_rng(4)
    w, h = 160, 120
    plane = np.array([0.02, 0.81, 0.586, -11.0]); plane[:3] /= np.linalg.norm(plane[:3])
    # points near the plane: x, y spread, z solved from the plane + a wave
    X = rng.uniform(-6, 6, (h, w)); Y = rng.uniform(-3, 3, (h, w))
    Z = (-plane[3] - plane[0] * X - plane[1] * Y) / plane[2] + 0.05 * np.sin(X * 2.0)
    valid = (rng.random((h, w)) < 0.7).astype(np.uint8)
    p3d = np.stack([X, Y, Z], axis=-1)
    mesh = gpu_ctx.mesh_upload(valid, p3d)
    args = dict(baseline=2.5, xmin=-12.0, xmax=12.0, ymin=-30.0, ymax=-5.0, width=96, height=80)
    grid, mask = mesh.grid_idw(plane, **args)
    pts = p3d[valid.astype(bool)].T.copy()
    ref, rmask = G.grid_idw(pts, plane, **args)
    np.testing.assert_array_equal(mask, rmask)
    assert 0.05 < mask.mean() < 1.0 and np.isnan(grid[mask == 0]).all()
    # Tolerance, derived, not tuned: the GPU accumulates z in 2^-24 fixed point (<= 3e-8 per point, hence per cell mean), the
    # 5x5 inverse-distance fill is a convex combination of cell means, and the output is float32 (relative 6e-8 of |z| <= 8 m
    # here = 5e-7).  The oracle is plain float64 and shares no arithmetic with the kernel.
    np.testing.assert_allclose(grid[mask == 1], ref[rmask == 1], rtol=0, atol=2e-6)
    # against the REFERENCE's cell statistic (nanmedian of ten random sub-samples, one seed): same mask; where a cell holds one
    # point the two statistics are the same number, elsewhere both lie inside the spread of the cell's points
    med, mmask = G.grid_idw(pts, plane, cell="subsample_median", seed=7, **args)
    np.testing.assert_array_equal(mask, mmask)
    R, T = G.compute_sea_plane_RT(plane)
    m = (R @ pts + T); m[2] *= -1.0; m *= args["baseline"]
    px = np.floor((m[0] - args["xmin"]) / (args["xmax"] - args["xmin"]) * (args["width"] - 1) + 0.5).astype(int)
    py = np.floor((m[1] - args["ymin"]) / (args["ymax"] - args["ymin"]) * (args["height"] - 1) + 0.5).astype(int)
    ok = (px >= 0) & (px < args["width"]) & (py >= 0) & (py < args["height"])
    lo = np.full((args["height"], args["width"]), np.inf); hi = -lo; cnt = np.zeros_like(lo)
    np.minimum.at(lo, (py[ok], px[ok]), m[2, ok]); np.maximum.at(hi, (py[ok], px[ok]), m[2, ok]); np.add.at(cnt, (py[ok], px[ok]), 1)
    single = cnt == 1
    assert single.sum() > 50
    np.testing.assert_allclose(grid[single], med[single], rtol=0, atol=2e-6)
    multi = cnt > 1
    assert (grid[multi] >= lo[multi] - 1e-6).all() and (grid[multi] <= hi[multi] + 1e-6).all()
    assert (med[multi] >= lo[multi] - 1e-6).all() and (med[multi] <= hi[multi] + 1e-6).all()
    spread = float(np.abs(grid[multi] - med[multi]).max())
    assert spread <= float((hi - lo)[multi].max())
    # the same cloud in a different point order gives the same grid, bit for bit (fixed-point accumulation)
    perm = rng.permutation(w * h)
    mesh2 = gpu_ctx.mesh_upload(valid.ravel()[perm].reshape(h, w), p3d.reshape(-1, 3)[perm].reshape(h, w, 3))
    grid2, _ = mesh2.grid_idw(plane, **args)
    np.testing.assert_array_equal(grid, grid2)


def test_grid_median_cells_match_oracle_and_resist_outliers(gpu_ctx):
    """wass_mesh_grid_idw_ex(WASS_GRID_CELL_MEDIAN): every cell holds the exact median of its points (np.median in the oracle),
    the grid does not depend on the point order, and a few gross outliers per cell move it far less than they move the mean --
    the property the reference's nanmedian of random sub-samples (wassgridsurface.py:330-345) has and a mean has not."""
    from oracle import grid_oracle as G
    rng = np.random.default_rng(11)
    w, h = 200, 150
    plane = np.array([0.02, 0.81, 0.586, -11.0]); plane[:3] /= np.linalg.norm(plane[:3])
    X = rng.uniform(-6, 6, (h, w)); Y = rng.uniform(-3, 3, (h, w))
    Z = (-plane[3] - plane[0] * X - plane[1] * Y) / plane[2] + 0.05 * np.sin(X * 2.0)
    out = rng.random((h, w)) < 0.02                              # 2 % spikes, 3 baselines off the surface
    Z = Z + out * 3.0
    valid = (rng.random((h, w)) < 0.8).astype(np.uint8)
    p3d = np.stack([X, Y, Z], axis=-1)
    mesh = gpu_ctx.mesh_upload(valid, p3d)
    args = dict(baseline=2.5, xmin=-12.0, xmax=12.0, ymin=-30.0, ymax=-5.0, width=48, height=40)     # coarse grid: many points per cell
    grid, mask = mesh.grid_idw(plane, cell="median", **args)
    pts = p3d[valid.astype(bool)].T.copy()
    ref, rmask = G.grid_idw(pts, plane, cell="median", **args)
    np.testing.assert_array_equal(mask, rmask)
    # float32 output of float64 medians (|z| <= 10 m: 1e-6), the inverse-distance fill a convex combination of them
    np.testing.assert_allclose(grid[mask == 1], ref[rmask == 1], rtol=0, atol=2e-6)
    # order independence, bit for bit
    perm = rng.permutation(w * h)
    mesh2 = gpu_ctx.mesh_upload(valid.ravel()[perm].reshape(h, w), p3d.reshape(-1, 3)[perm].reshape(h, w, 3))
    grid2, _ = mesh2.grid_idw(plane, cell="median", **args)
    np.testing.assert_array_equal(grid, grid2)
    # robustness: against the clean surface (no spikes) the median grid is several times closer than the mean grid
    clean = np.stack([X, Y, Z - out * 3.0], axis=-1)
    truth, _ = G.grid_idw(clean[valid.astype(bool)].T.copy(), plane, cell="median", **args)
    mean_grid, _ = mesh.grid_idw(plane, cell="mean", **args)
    e_med = np.abs(grid[mask == 1] - truth[mask == 1]).mean(); e_mean = np.abs(mean_grid[mask == 1] - truth[mask == 1]).mean()
    assert e_med < 0.25 * e_mean, (e_med, e_mean)


def test_grid_of_the_synthetic_sea_plane_is_flat(gpu_ctx):
    """Whole path: SGM -> clean-up -> triangulation -> plane fit -> grid.  Aligned on its own plane the synthetic surface
    (an exact plane plus a +-0.6 px disparity ripple) is flat to a few centimetres at a 2.5 m baseline."""
    w, h, D = 640, 480, 64
    right, left = synth.make_pair(w, h, D, frame_idx=3)
    p = default_sgm_params(D, ndirs=5)
    roi = (0, 0, w, h)
    f = gpu_ctx.disparity_postprocess(gpu_ctx.sgm_disparity(right, left, p), p)
    mesh, n = gpu_ctx.triangulate(f, w, h, roi, roi, wass_amd.make_geom(synth.rig_geometry(w, h)), right, None, (right <= 254).astype(np.uint8))
    mesh.remove_outliers(99.0)
    res = mesh.fit_plane(wass_amd.ransac_sample(w, h, 400, 12345), 1.0, 1.5)
    assert res.found
    plane = np.array(res.plane[:])
    grid, mask = mesh.grid_idw(plane, 2.5, -20.0, 20.0, -80.0, -20.0, 200, 300)
    assert mask.mean() > 0.2
    z = grid[mask == 1]
    assert abs(np.nanmean(z)) < 0.05 and np.nanstd(z) < 0.5
