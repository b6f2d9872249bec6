"""GPU parity for row f1 (rectification resamplers) against the CPU oracle, through the C ABI: bit-exact
(integer fixed-point pipelines)."""
import numpy as np
import pytest

import wass_amd
from wass_amd import synth

pytestmark = pytest.mark.gpu


def _rot(v):
    v = np.asarray(v, float); th = np.linalg.norm(v)
    k = v / th
    Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * Kx @ Kx


def _maps(rng, sw, sh, dw, dh, spill):
    u, v = np.meshgrid(np.arange(dw, dtype=np.float64), np.arange(dh, dtype=np.float64))
    a = rng.normal(0, 0.02, 4)
    mx = (u * (sw / dw) * (1 + a[0]) + v * a[1] + rng.uniform(-spill, spill) + 2.0 * np.sin(v / 17.0)).astype(np.float32)
    my = (v * (sh / dh) * (1 + a[2]) + u * a[3] + rng.uniform(-spill, spill) + 1.5 * np.cos(u / 23.0)).astype(np.float32)
    return mx, my


@pytest.mark.parametrize("sw,sh,dw,dh,spill", [(200, 150, 200, 150, 6.0), (97, 61, 130, 40, 3.0), (64, 48, 64, 48, 80.0),
                                               (5, 4, 33, 17, 2.0), (3, 3, 20, 20, 1.0), (1, 1, 8, 8, 1.0)])
def test_remap_cubic_parity(gpu_ctx, oracle, sw, sh, dw, dh, spill):
    rng = np.random.default_rng(sw * 1000 + dh)
    img = rng.integers(0, 256, (sh, sw), dtype=np.uint8)
    mx, my = _maps(rng, sw, sh, dw, dh, spill)
    ref = oracle.remap_cubic(img, mx, my)
    got = gpu_ctx.remap_cubic(img, mx, my)
    np.testing.assert_array_equal(got, ref)
    if dw > 10 and dh > 10:
        roi = (3, 2, dw - 7, dh - 5)
        np.testing.assert_array_equal(gpu_ctx.remap_cubic(img, mx, my, roi=roi), ref[2:2 + dh - 5, 3:3 + dw - 7])


def test_remap_cubic_extreme_coordinates(gpu_ctx, oracle):
    """maps far outside / huge / negative: saturate_cast<short> of the integer part, all-border pixels are 0"""
    rng = np.random.default_rng(8)
    img = rng.integers(1, 256, (40, 50), dtype=np.uint8)
    mx = rng.uniform(-3, 53, (30, 60)).astype(np.float32); my = rng.uniform(-3, 43, (30, 60)).astype(np.float32)
    mx[0, :10] = [-1e6, 1e6, -40000, 40000, -1.5, -2.96875, 49.0, 50.96875, 51.0, 3e9]
    my[1, :6] = [-1e6, 1e6, -2.96875, 41.0, 39.5, -1.0]
    np.testing.assert_array_equal(gpu_ctx.remap_cubic(img, mx, my), oracle.remap_cubic(img, mx, my))


@pytest.mark.parametrize("sw,sh,dw,dh", [(200, 150, 200, 150), (97, 61, 130, 40), (80, 60, 50, 9), (300, 20, 300, 33), (2, 2, 40, 30)])
def test_warp_perspective_parity(gpu_ctx, oracle, sw, sh, dw, dh):
    rng = np.random.default_rng(sw + 7 * dh)
    img = rng.integers(0, 256, (sh, sw), dtype=np.uint8)
    for trial in range(4):
        Hm = np.eye(3) + rng.normal(0, 0.03, (3, 3))
        Hm[0, 2] = rng.uniform(-8, 8); Hm[1, 2] = rng.uniform(-8, 8)
        Hm[2, 0] = rng.normal(0, 2e-4); Hm[2, 1] = rng.normal(0, 2e-4); Hm[2, 2] = 1 + rng.normal(0, 0.01)
        Hm = np.diag([dw / sw, dh / sh, 1.0]) @ Hm
        ref = oracle.warp_perspective(img, Hm, dw, dh)
        np.testing.assert_array_equal(gpu_ctx.warp_perspective(img, Hm, dw, dh), ref)
        if dw > 20 and dh > 20:
            roi = (5, 4, dw - 11, dh - 9)
            np.testing.assert_array_equal(gpu_ctx.warp_perspective(img, Hm, dw, dh, roi=roi), ref[4:4 + dh - 9, 5:5 + dw - 11])
    np.testing.assert_array_equal(gpu_ctx.warp_perspective(img, np.eye(3), sw, sh), img)


@pytest.mark.parametrize("w,h", [(90, 70), (333, 41), (2049, 9), (5000, 5)])
def test_undistort_parity(gpu_ctx, oracle, w, h):
    """Row f2: cv::undistort (wass_prepare.cpp:268), bit-exact against the oracle; widths on both sides of the
    4096-pixel stripe rule (several rows per stripe / one row per stripe)."""
    rng = np.random.default_rng(w)
    img = rng.integers(0, 256, (h, w), dtype=np.uint8)
    K = np.array([[0.9 * w, 0, w / 2 - 0.3], [0, 0.92 * w, h / 2 + 0.2], [0, 0, 1]])
    for dist in ([0, 0, 0, 0, 0], [-0.25, 0.08, 0.002, -0.001, 0.01], [-0.2, 0.05, 0.001, -0.002],
                 [0.1, -0.02, 0, 0, 0.003, 0.01, -0.005, 0.001],
                 [-0.2, 0.05, 0.001, -0.002, 0.01, 0.0, 0.0, 0.0, 0.002, -0.001, 0.0015, 0.0005], [3.0, -8.0, 0.1, 0.1, 5.0]):
        np.testing.assert_array_equal(gpu_ctx.undistort(img, K, dist), oracle.undistort(img, K, dist))
    with pytest.raises(wass_amd.WassError):
        gpu_ctx.undistort(img, K, [0.1] * 14)


def test_undistort_fullsize(gpu_ctx, oracle):
    w, h = 2456, 2058
    img = synth.make_pair(w, h, 256, frame_idx=2)[0]
    K = synth.rig_geometry(w, h)["K_left"]
    dist = [-0.12, 0.03, 0.0005, -0.0007, 0.002]
    np.testing.assert_array_equal(gpu_ctx.undistort(img, K, dist), oracle.undistort(img, K, dist))


def test_resampler_argument_errors(gpu_ctx):
    img = np.zeros((10, 12), np.uint8)
    with pytest.raises(wass_amd.WassError):
        gpu_ctx.warp_perspective(img, np.zeros((3, 3)), 12, 10)                     # singular homography
    with pytest.raises(wass_amd.WassError):
        gpu_ctx.warp_perspective(img, np.eye(3), 12, 10, roi=(4, 4, 12, 10))        # roi outside the destination
    mx = np.zeros((10, 12), np.float32)
    with pytest.raises(wass_amd.WassError):
        gpu_ctx.remap_cubic(img, mx, mx, roi=(0, 0, 0, 5))


def test_device_resident_variants_and_stride(gpu_ctx, oracle):
    import torch
    rng = np.random.default_rng(3)
    big = rng.integers(0, 256, (90, 160), dtype=np.uint8)
    view = big[:, 10:130]                                                           # stride 160, width 120
    d_big = torch.from_numpy(big).cuda()
    d_view = d_big[:, 10:130]
    mx, my = _maps(rng, 120, 90, 100, 70, 4.0)
    ref = oracle.remap_cubic(np.ascontiguousarray(view), mx, my)
    roi = (7, 3, 80, 60)
    got = gpu_ctx.remap_cubic_dev(d_view, torch.from_numpy(mx).cuda(), torch.from_numpy(my).cuda(), roi=roi)
    gpu_ctx.synchronize()
    np.testing.assert_array_equal(got.cpu().numpy(), ref[3:63, 7:87])
    Hm = np.array([[1.01, 0.02, 3.0], [-0.015, 0.99, -2.0], [1e-4, -5e-5, 1.0]])
    got = gpu_ctx.warp_perspective_dev(d_view, Hm, 120, 90)
    gpu_ctx.synchronize()
    np.testing.assert_array_equal(got.cpu().numpy(), oracle.warp_perspective(np.ascontiguousarray(view), Hm, 120, 90))


def test_fullsize_opencv_rectification_config_b(gpu_ctx, oracle):
    """BASELINE config B frame through the cv::stereoRectify path of a slightly convergent rig: host maps equal the
    oracle's, the GPU bicubic remap of the full 2456x2058 frame is bit-exact, the fused ROI crop equals slicing."""
    w, h = 2456, 2058
    right, left = synth.make_pair(w, h, 256, frame_idx=5)
    K = synth.rig_geometry(w, h)["K_left"]
    R = _rot([0.004, -0.02, 0.003]); T = np.array([0.999, 0.01, -0.03])
    rr = wass_amd.stereo_rectify(K, K, w, h, R, T, 1.0)
    ro = oracle.stereo_rectify(K, K, w, h, R, T, 1.0)
    for k in rr:
        np.testing.assert_array_equal(np.array(rr[k]), np.array(ro[k]))
    mx, my = wass_amd.init_rectify_map(K, rr["R1"], rr["P1"], w, h)
    ox, oy = oracle.init_rectify_map(K, rr["R1"], rr["P1"], w, h)
    np.testing.assert_array_equal(mx, ox); np.testing.assert_array_equal(my, oy)
    ref = oracle.remap_cubic(left, mx, my)
    np.testing.assert_array_equal(gpu_ctx.remap_cubic(left, mx, my), ref)
    x, y, rw, rh = rr["roi1"]
    assert rw > 0.8 * w and rh > 0.8 * h
    np.testing.assert_array_equal(gpu_ctx.remap_cubic(left, mx, my, roi=rr["roi1"]), ref[y:y + rh, x:x + rw])
    assert ref[y:y + rh, x:x + rw].min() >= 0 and (ref[y + 2:y + rh - 2, x + 2:x + rw - 2] > 0).mean() > 0.99    # inside the valid ROI
