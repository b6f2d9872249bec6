"""Row f1 (rectification): the oracle restatement of cv::stereoRectify / initUndistortRectifyMap / remap /
warpPerspective against independent float-math properties, and the host math of libwassgpu against the oracle.

OpenCV is absent from this image, so these are property tests (PARITY UNPINNED): epipolar alignment, map
consistency with the projective model, identity / integer-shift resampling, and agreement of the fixed-point
bicubic with a float64 Keys (a = -0.75) interpolation to within one grey level.
"""
import numpy as np
import pytest

import wass_amd

K1 = np.array([[2000., 0, 1227.5], [0, 2000., 1028.5], [0, 0, 1]])
K2 = np.array([[2010., 0, 1200.5], [0, 2005., 1040.5], [0, 0, 1]])
W, H = 2456, 2058


def _rot(v):
    v = np.asarray(v, float); th = np.linalg.norm(v)
    if th == 0:
        return np.eye(3)
    k = v / th
    Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * Kx @ Kx


def _rigs(n=12, seed=3):
    rng = np.random.default_rng(seed)
    for _ in range(n):
        yield (_rot(rng.normal(0, 0.03, 3)), np.array([rng.choice([-1., 1.]), 0, 0]) + rng.normal(0, 0.05, 3),
               float(rng.choice([-1.0, 0.0, 0.5, 1.0])))


def test_inter_tables(oracle):
    for ks in (2, 4):
        t = oracle.inter_tab(ks).astype(np.int64)
        assert (t.sum(axis=(1, 2)) == 32768).all()            # every entry sums to INTER_REMAP_COEF_SCALE
    lin = oracle.inter_tab(2)
    assert lin[0].tolist() == [[32767, 0], [0, 1]]            # saturate_cast<short>(32768) + sum fix-up
    assert lin[16 * 32 + 16].tolist() == [[8192, 8192], [8192, 8192]]
    cub = oracle.inter_tab(4)
    assert cub[0, 1, 1] == 32767 and cub[0].sum() == 32768
    # separable up to rounding: centre-phase weights are the Keys kernel at +-0.5, +-1.5
    assert cub[16 * 32 + 16, 1, 1] == 11552 and cub[16 * 32 + 16, 0, 0] == 288


def test_stereo_rectify_epipolar_geometry(oracle):
    rng = np.random.default_rng(0)
    for R, T, alpha in _rigs():
        r = oracle.stereo_rectify(K1, K2, W, H, R, T, alpha)
        for Rk in (r["R1"], r["R2"]):
            np.testing.assert_allclose(Rk @ Rk.T, np.eye(3), atol=1e-12)
            assert abs(np.linalg.det(Rk) - 1) < 1e-12
        # x2 = R x1 + T: a point seen by both rectified cameras lands on the same row
        X1 = np.c_[rng.uniform(-5, 5, 50), rng.uniform(-5, 5, 50), rng.uniform(20, 60, 50)].T
        X2 = R @ X1 + T[:, None]
        p1 = r["P1"][:, :3] @ (r["R1"] @ X1)
        p2 = r["P2"][:, :3] @ (r["R2"] @ X2)
        np.testing.assert_allclose(p1[1] / p1[2], p2[1] / p2[2], atol=1e-8)
        # ... and P2 applied to rectified-camera-1 coordinates gives the same pixel as the second camera
        q2 = r["P2"] @ np.vstack([r["R1"] @ X1, np.ones(50)])
        np.testing.assert_allclose(q2[:2] / q2[2], p2[:2] / p2[2], atol=1e-7)
        assert r["P1"][0, 0] == r["P1"][1, 1] == r["P2"][0, 0] == r["P2"][1, 1]
        assert r["P1"][1, 2] == r["P2"][1, 2]                                   # horizontal stereo: common cy
        for roi in (r["roi1"], r["roi2"]):
            assert roi[0] >= 0 and roi[1] >= 0 and roi[0] + roi[2] <= W and roi[1] + roi[3] <= H


def test_stereo_rectify_ideal_rig(oracle):
    """even the ideal rig is not a pass-through ((nx-1)/2 integer centre, [0,W-1] grid)."""
    K = K1
    r = oracle.stereo_rectify(K, K, W, H, np.eye(3), [1.0, 0, 0], 1.0)
    np.testing.assert_array_equal(r["R1"], np.eye(3)); np.testing.assert_array_equal(r["R2"], np.eye(3))
    assert r["P1"][0, 2] == (W - 1) // 2 and r["P1"][1, 2] == (H - 1) // 2
    assert 0.999 < r["P1"][0, 0] / K[0, 0] < 1.0
    assert r["P2"][0, 3] == r["P2"][0, 0] * 1.0
    assert r["roi1"] == r["roi2"] and r["roi1"][2] >= W - 4 and r["roi1"][3] >= H - 3


def test_host_stereo_rectify_matches_oracle(oracle):
    for R, T, alpha in _rigs(20, seed=9):
        a = oracle.stereo_rectify(K1, K2, W, H, R, T, alpha)
        b = wass_amd.stereo_rectify(K1, K2, W, H, R, T, alpha)
        for k in a:
            np.testing.assert_array_equal(np.array(a[k]), np.array(b[k]), err_msg=k)
    with pytest.raises(wass_amd.WassError):
        wass_amd.stereo_rectify(K1, K2, W, H, np.eye(3), [0, 0, 0])


def test_init_rectify_map(oracle):
    R, T, _ = next(_rigs(1, seed=4))
    r = oracle.stereo_rectify(K1, K2, 640, 480, R, T, 1.0)
    for K, Rk, P in ((K1, r["R1"], r["P1"]), (K2, r["R2"], r["P2"])):
        mx, my = oracle.init_rectify_map(K, Rk, P, 640, 480)
        hx, hy = wass_amd.init_rectify_map(K, Rk, P, 640, 480)
        np.testing.assert_array_equal(mx, hx); np.testing.assert_array_equal(my, hy)
        # rectified pixel (u,v) -> ray P^-1 (u,v,1) -> back-rotate by R^T -> source pixel through K
        u, v = np.meshgrid(np.arange(640.), np.arange(480.))
        ray = Rk.T @ np.linalg.inv(P[:, :3]) @ np.stack([u.ravel(), v.ravel(), np.ones(u.size)])
        src = K @ (ray / ray[2])
        np.testing.assert_allclose(mx.ravel(), src[0], atol=2e-4)
        np.testing.assert_allclose(my.ravel(), src[1], atol=2e-4)


def _keys(x, a=-0.75):
    x = np.abs(x)
    return np.where(x <= 1, (a + 2) * x**3 - (a + 3) * x**2 + 1, np.where(x < 2, a * x**3 - 5 * a * x**2 + 8 * a * x - 4 * a, 0.0))


def test_oracle_remap_cubic_properties(oracle):
    rng = np.random.default_rng(2)
    img = rng.integers(0, 256, (60, 80), dtype=np.uint8)
    u, v = np.meshgrid(np.arange(80, dtype=np.float32), np.arange(60, dtype=np.float32))
    np.testing.assert_array_equal(oracle.remap_cubic(img, u, v), img)                      # identity
    sh = oracle.remap_cubic(img, u + 3, v - 2)                                             # integer shift, zero border
    ref = np.zeros_like(img); ref[2:, :-3] = img[:-2, 3:]
    np.testing.assert_array_equal(sh, ref)
    # smooth image, fractional map: fixed point vs float64 Keys interpolation at the 1/32-quantised position
    yy, xx = np.mgrid[0:60, 0:80]
    smooth = (127 + 60 * np.sin(xx / 7.0) + 50 * np.cos(yy / 5.0)).astype(np.uint8)
    mx = (u * 0.93 + 2.37).astype(np.float32); my = (v * 0.95 + 0.011 * u + 1.61).astype(np.float32)
    got = oracle.remap_cubic(smooth, mx, my).astype(np.int64)
    qx = np.rint(mx.astype(np.float64) * 32) / 32; qy = np.rint(my.astype(np.float64) * 32) / 32
    ix = np.floor(qx).astype(int); iy = np.floor(qy).astype(int)
    P = np.pad(smooth.astype(np.float64), 4)
    acc = np.zeros_like(qx)
    for i in range(-1, 3):
        for j in range(-1, 3):
            acc += _keys(qy - (iy + i)) * _keys(qx - (ix + j)) * P[np.clip(iy + i + 4, 0, 67), np.clip(ix + j + 4, 0, 87)]
    inside = (ix >= 1) & (ix < 77) & (iy >= 1) & (iy < 57)
    assert np.abs(got - np.clip(np.rint(acc), 0, 255))[inside].max() <= 1


def test_oracle_warp_perspective_properties(oracle):
    rng = np.random.default_rng(4)
    img = rng.integers(0, 256, (50, 90), dtype=np.uint8)
    np.testing.assert_array_equal(oracle.warp_perspective(img, np.eye(3), 90, 50), img)
    Ht = np.array([[1, 0, 5.0], [0, 1, -3.0], [0, 0, 1]])          # dst(x,y) = src(x-5, y+3)
    ref = np.zeros_like(img); ref[:-3, 5:] = img[3:, :-5]
    np.testing.assert_array_equal(oracle.warp_perspective(img, Ht, 90, 50), ref)
    Hs = np.array([[1, 0, 0.5], [0, 1, 0], [0, 0, 1]])             # half-pixel shift = mean of neighbours (rounded)
    got = oracle.warp_perspective(img, Hs, 90, 50).astype(int)
    exp = (img[:, :-1].astype(int) + img[:, 1:].astype(int) + 1) >> 1
    assert np.abs(got[:, 1:] - exp).max() <= 1


def test_oracle_undistort_properties(oracle):
    """Row f2: cv::undistort of wass_prepare.cpp:268 (restated, unpinned): zero distortion is the identity; a
    distorted smooth image agrees with a float64 evaluation of the Brown model + bilinear interpolation at the
    1/32-quantised position to within one grey level."""
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, (70, 90), dtype=np.uint8)
    K = np.array([[80., 0, 44.5], [0, 82., 35.5], [0, 0, 1]])
    np.testing.assert_array_equal(oracle.undistort(img, K, [0, 0, 0, 0, 0]), img)
    yy, xx = np.mgrid[0:70, 0:90]
    smooth = (127 + 60 * np.sin(xx / 6.0) + 50 * np.cos(yy / 5.0)).astype(np.uint8)
    for dist in ([-0.25, 0.08, 0.002, -0.001, 0.01], [-0.2, 0.05, 0.001, -0.002], [0.1, -0.02, 0, 0, 0.003, 0.01, -0.005, 0.001],
                 [-0.2, 0.05, 0.001, -0.002, 0.01, 0.0, 0.0, 0.0, 0.002, -0.001, 0.0015, 0.0005]):
        got = oracle.undistort(smooth, K, dist).astype(np.int64)
        k = list(dist) + [0.0] * (12 - len(dist))
        x = (xx - K[0, 2]) / K[0, 0]; y = (yy - K[1, 2]) / K[1, 1]
        r2 = x * x + y * y
        kr = (1 + ((k[4] * r2 + k[1]) * r2 + k[0]) * r2) / (1 + ((k[7] * r2 + k[6]) * r2 + k[5]) * r2)
        xd = x * kr + k[2] * 2 * x * y + k[3] * (r2 + 2 * x * x) + k[8] * r2 + k[9] * r2 * r2
        yd = y * kr + k[2] * (r2 + 2 * y * y) + k[3] * 2 * x * y + k[10] * r2 + k[11] * r2 * r2
        u = np.rint((K[0, 0] * xd + K[0, 2]) * 32) / 32; v = np.rint((K[1, 1] * yd + K[1, 2]) * 32) / 32
        iu = np.floor(u).astype(int); iv = np.floor(v).astype(int); fu = u - iu; fv = v - iv
        P = np.pad(smooth.astype(np.float64), 2)
        at = lambda r, c: P[np.clip(r + 2, 0, 73), np.clip(c + 2, 0, 93)]  # noqa: E731
        ref = (1 - fv) * ((1 - fu) * at(iv, iu) + fu * at(iv, iu + 1)) + fv * ((1 - fu) * at(iv + 1, iu) + fu * at(iv + 1, iu + 1))
        inside = (iu >= 0) & (iu < 89) & (iv >= 0) & (iv < 69)
        assert inside.mean() > 0.8
        assert np.abs(got - np.rint(ref))[inside].max() <= 1
    with pytest.raises(ValueError):
        oracle.undistort(img, K, [0.1] * 14)                      # tilt model is not restated
