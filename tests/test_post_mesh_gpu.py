"""GPU parity for SURVEY.md section 8 rows a7-a20 against the CPU oracle, through the C ABI.

Bit-exact where the arithmetic is integer / float32 with a fixed operation order / exact fp64 comparisons
(filters, validity masks, inlier counts, order statistics, xyzC bytes); fp64 sums whose order differs on the
GPU (plane refinement) are compared with the tolerance written in the test.
"""
import struct

import numpy as np
import pytest

import wass_amd
from wass_amd import default_sgm_params, synth

pytestmark = pytest.mark.gpu


def _oracle_params(O, p):
    return O.SgbmParams(p.min_disp, p.num_disp, p.win, p.P1, p.P2, p.uniq_ratio, p.disp12_max_diff,
                        p.prefilter_cap, p.speckle_win, p.speckle_range, p.ndirs)


# ------------------------------------------------------------------ a7-a9
@pytest.mark.parametrize("w,h,D,dil,ero,off", [(160, 120, 32, 1, 2, 0), (131, 77, 48, 2, 1, 0), (131, 77, 48, 0, 0, 3),
                                               (97, 64, 32, 3, 3, -2), (20, 3, 16, 1, 2, 0), (40, 2, 16, 1, 1, 0)])
def test_postprocess_parity(gpu_ctx, oracle, w, h, D, dil, ero, off):
    right, left = synth.make_pair(w, h, D, frame_idx=7 * w + h)
    p = default_sgm_params(D, ndirs=5, disp_offset=off)
    d16, _ = oracle.dense_disparity16(right, left, _oracle_params(oracle, p), off)
    # punch holes so that dilate/erode have work to do
    rng = np.random.default_rng(w)
    d16 = d16.copy(); d16[rng.random(d16.shape) < 0.05] = 0
    got = gpu_ctx.disparity_postprocess(d16, p, dil, ero, 0)
    ref = oracle.disparity_postprocess(d16, p.min_disp, D, max(off, 0), dil, ero)
    np.testing.assert_array_equal(got, ref)


def test_postprocess_one_kernel_per_step_path(gpu_ctx, oracle, monkeypatch):
    """WASS_CLEAN_CHAIN=0 selects the one-kernel-per-step path (what DENSE_SCALE != 1 always uses) instead of the fused tile kernel."""
    rng = np.random.default_rng(11)
    d16 = (rng.integers(20, 400, (75, 210))).astype(np.int16)
    d16[rng.random(d16.shape) < 0.1] = 0
    p = default_sgm_params(32)
    fused = gpu_ctx.disparity_postprocess(d16, p, 1, 2, 3)
    monkeypatch.setenv("WASS_CLEAN_CHAIN", "0")
    steps = gpu_ctx.disparity_postprocess(d16, p, 1, 2, 3)
    np.testing.assert_array_equal(fused, steps)
    np.testing.assert_array_equal(gpu_ctx.disparity_postprocess(d16, p, 2, 1, 0), oracle.disparity_postprocess(d16, 1, 32, 0, 2, 1))


def test_postprocess_stage_by_stage(gpu_ctx, oracle):
    """dilate only / erode only, against the oracle's single filters (the mask step adds one erosion)."""
    rng = np.random.default_rng(5)
    d16 = (rng.integers(20, 400, (60, 90))).astype(np.int16)
    d16[rng.random(d16.shape) < 0.2] = 0
    p = default_sgm_params(32)
    f = oracle.clean_and_convert(d16, 1, 32)
    got = gpu_ctx.disparity_postprocess(d16, p, 1, 0, 0)
    np.testing.assert_array_equal(got, oracle.erode_zero(oracle.dilate_zero(f)))
    got = gpu_ctx.disparity_postprocess(d16, p, 0, 1, 0)
    np.testing.assert_array_equal(got, oracle.erode_zero(oracle.erode_zero(f)))


@pytest.mark.parametrize("ks", [3, 5])
def test_postprocess_median(gpu_ctx, oracle, ks):
    rng = np.random.default_rng(9)
    d16 = (rng.integers(20, 400, (50, 70))).astype(np.int16)
    p = default_sgm_params(32)
    base = oracle.disparity_postprocess(d16, 1, 32, 0, 0, 0)
    got = gpu_ctx.disparity_postprocess(d16, p, 0, 0, ks)
    r = ks // 2
    pad = np.pad(base, r, mode="edge")
    win = np.stack([pad[i:i + 50, j:j + 70] for i in range(ks) for j in range(ks)], 0)
    np.testing.assert_array_equal(got, np.median(win, axis=0).astype(np.float32))


def test_postprocess_device_entry(gpu_ctx, oracle):
    import torch
    rng = np.random.default_rng(2)
    d16 = (rng.integers(0, 600, (64, 80))).astype(np.int16)
    p = default_sgm_params(32)
    out = gpu_ctx.disparity_postprocess_dev(torch.from_numpy(d16).cuda(), p)
    gpu_ctx.synchronize()
    np.testing.assert_array_equal(out.cpu().numpy(), oracle.disparity_postprocess(d16, 1, 32))


# ------------------------------------------------------------------ a10-a13
def _scene(w=200, h=150, D=48, frame=11):
    right, left = synth.make_pair(w, h, D, frame_idx=frame)
    gt = synth.true_disparity(w, h, D).astype(np.float32)
    rng = np.random.default_rng(frame)
    disp = gt.copy()
    disp[rng.random(disp.shape) < 0.1] = 0          # holes
    disp[:, :3] = 0.5                               # below min_disp
    return right, left, disp


@pytest.mark.parametrize("use_custom", [False, True])
def test_triangulate_parity(gpu_ctx, oracle, use_custom):
    w, h, D = 200, 150, 48
    right, left, disp = _scene(w, h, D)
    rig = synth.rig_geometry(w, h)
    if use_custom:                                   # a non-trivial homography pair
        Hl = np.array([[1.01, 0.002, -1.5], [-0.001, 0.995, 0.7], [1e-6, -2e-6, 1.0]])
        Hr = np.array([[0.995, -0.001, 0.8], [0.0015, 1.005, -0.4], [-1e-6, 1e-6, 1.0]])
        rig["HLi"] = np.linalg.inv(Hl); rig["HRi"] = np.linalg.inv(Hr)
    else:                                            # a small rectifying rotation
        a = 0.01
        Rr = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1.0]])
        rig["R1"] = Rr; rig["R2"] = Rr.T
    roi_l, roi_r = (2, 1, w - 4, h - 3), (3, 1, w - 4, h - 3)
    droi = np.ascontiguousarray(disp[roi_r[1]:roi_r[1] + roi_r[3], roi_r[0]:roi_r[0] + roi_r[2]])
    full = np.zeros((h, w), np.float32); full[roi_r[1]:roi_r[1] + roi_r[3], roi_r[0]:roi_r[0] + roi_r[2]] = droi
    lmask = (np.random.default_rng(1).random((h, w)) > 0.05).astype(np.uint8)
    rmask = (right <= 254).astype(np.uint8)
    bbox = (5.0, 4.0, w - 6.0, h - 5.0)
    og = oracle.make_geom(rig, use_custom=use_custom, disparity_compensation=1.0)
    n_ref, v_ref, p_ref, g_ref = oracle.triangulate(full, roi_l, roi_r, og, right, lmask, rmask, 20.0, bbox, 1.0)
    gg = wass_amd.make_geom(rig, use_custom=use_custom, disparity_compensation=1.0)
    mesh, n = gpu_ctx.triangulate(droi, w, h, roi_l, roi_r, gg, right, lmask, rmask, 20.0, bbox, 1.0)
    valid, p3d, gray = mesh.download()
    assert n_ref > 1000
    # acos() differs in the last bits between glibc and the device: allow a handful of threshold flips
    flips = int((valid != v_ref).sum())
    assert flips <= 2 and abs(n - n_ref) <= 2
    both = (valid == 1) & (v_ref == 1)
    np.testing.assert_allclose(p3d[both], p_ref[both], rtol=1e-12, atol=1e-12)   # fp64, same op order
    np.testing.assert_array_equal(gray[both], g_ref[both])
    assert (p3d[valid == 0] == 0).all()
    # what the reference paints into undistorted/R0.jpg / R1.jpg (wass_stereo.cpp:1216-1338): grey exactly where a point was
    # made, black where there was no disparity, yellow in R0 only where only the LEFT mask rejected the pixel
    c0, c1 = mesh.reject_codes()
    GREY, NONE, BBOX = 1, 0, 3
    np.testing.assert_array_equal((c0 == GREY) & (c1 == GREY), valid == 1)
    assert ((c0 == NONE) == (c1 == NONE)).all() and (c0[droi <= 1.0] == NONE).all() and (c0 == NONE).sum() < 0.5 * c0.size
    left_only = (c0 == BBOX) & (c1 == GREY)
    assert 0.01 * c0.size < left_only.sum() < 0.12 * c0.size          # the left mask drops 5 % of the pixels at random
    assert set(np.unique(c0)) <= {0, 1, 2, 3, 4, 5, 6} and set(np.unique(c1)) <= {0, 1, 2, 3, 4, 5, 6}


def test_triangulate_depth_matches_ground_truth(gpu_ctx):
    """Z = f*B/d for the ideal rig (B = 1): ties the geometry to the synthetic scene."""
    w, h, D = 160, 120, 32
    right, left = synth.make_pair(w, h, D)
    gt = synth.true_disparity(w, h, D).astype(np.float32)
    rig = synth.rig_geometry(w, h)
    mesh, n = gpu_ctx.triangulate(gt, w, h, (0, 0, w, h), (0, 0, w, h), wass_amd.make_geom(rig), right)
    valid, p3d, _ = mesh.download()
    m = valid == 1
    assert m.mean() > 0.5
    np.testing.assert_allclose(p3d[..., 2][m], (0.9 * w) / gt[m], rtol=1e-5)


# ------------------------------------------------------------------ a14-a20
def _cloud(w=210, h=140, seed=0, holes=0.15, outliers=0.03):
    rng = np.random.default_rng(seed)
    n = np.array([0.04, -0.45, 0.89]); n /= np.linalg.norm(n)
    d = -18.0
    u, v = np.meshgrid(np.arange(w), np.arange(h))
    x = (u - w / 2) * 0.12 + rng.normal(0, 0.01, (h, w))
    y = (v - h / 2) * 0.1 + rng.normal(0, 0.01, (h, w))
    z = (-d - n[0] * x - n[1] * y) / n[2] + rng.normal(0, 0.15, (h, w))
    o = rng.random((h, w)) < outliers
    z[o] += rng.uniform(3, 30, o.sum())
    valid = (rng.random((h, w)) > holes).astype(np.uint8)
    valid[60:64, :] = 0; valid[60:64, 100:103] = 1          # a thin bridge between two big parts
    p3d = np.ascontiguousarray(np.stack([x, y, z], -1))
    p3d[valid == 0] = 0
    return valid, p3d, np.array([*n, d])


def test_mesh_roundtrip(gpu_ctx):
    valid, p3d, _ = _cloud()
    gray = (np.arange(valid.size) % 251).astype(np.uint8).reshape(valid.shape)
    m = gpu_ctx.mesh_upload(valid, p3d, gray)
    v2, p2, g2 = m.download()
    np.testing.assert_array_equal(v2, valid); np.testing.assert_array_equal(p2, p3d); np.testing.assert_array_equal(g2, gray)


@pytest.mark.parametrize("pct", [99.0, 50.0, 0.0, 100.0, 98.7])
def test_zgap_percentile_exact(gpu_ctx, oracle, pct):
    valid, p3d, _ = _cloud(seed=3)
    m = gpu_ctx.mesh_upload(valid, p3d)
    got, n = m.zgap_percentile(pct)
    ref, n_ref = oracle.zgap_percentile(valid, p3d, pct)
    assert n == n_ref and got == ref


def _zgap_ref(valid, z, pct):
    """PovMesh.cpp:888-926 in numpy: |z - z(neighbour k in the row above)|, k = -1, 0, +1, columns 1 .. w-2, rows 1 .. h-1; sorted; [floor(p/100 n)]"""
    v = valid.astype(bool)
    g = []
    for k in (-1, 0, 1):
        c = v[1:, 1:-1] & v[:-1, 1 + k:z.shape[1] - 1 + k]
        g.append(np.abs(z[1:, 1:-1] - z[:-1, 1 + k:z.shape[1] - 1 + k])[c])
    g = np.sort(np.concatenate(g))
    if g.size == 0:
        return float("nan"), 0
    return float(g[min(int(np.floor(pct / 100.0 * g.size)), g.size - 1)]), int(g.size)


@pytest.mark.parametrize("pct", [99.0, 50.0, 0.0, 100.0, 3.0])
def test_frame_tail_zgap_is_the_exact_order_statistic_on_rough_and_on_quantised_surfaces(gpu_ctx, pct):
    """The device-side select of the frame tail (wass_mesh_remove_outliers) against numpy: a rough surface (1.4 M distinct gaps) and a
    quantised one (millions of gaps share a handful of values: every radix pass but the first sees ties only)."""
    rng = np.random.default_rng(17)
    h, w = 600, 800
    rough = np.cumsum(rng.normal(0, 0.05, (h, w)), axis=0) + rng.normal(0, 0.3, (h, w)) ** 3
    quant = np.cumsum(rng.integers(0, 3, (h, w)).astype(np.float64) * 0.25, axis=0)
    valid = (rng.random((h, w)) < 0.93).astype(np.uint8)
    for z in (rough, quant):
        p3d = np.zeros((h, w, 3)); p3d[..., 2] = z
        m = gpu_ctx.mesh_upload(valid, p3d)
        got, n, _ = m.remove_outliers(pct)
        assert (got, n) == _zgap_ref(valid, z, pct)


def test_zgap_empty(gpu_ctx):
    valid = np.zeros((10, 12), np.uint8)
    m = gpu_ctx.mesh_upload(valid, np.zeros((10, 12, 3)))
    got, n = m.zgap_percentile(99.0)
    assert n == 0 and np.isnan(got)


@pytest.mark.parametrize("seed,gapscale", [(0, 1.0), (1, 0.5), (2, 0.2), (3, 3.0)])
def test_biggest_component_exact(gpu_ctx, oracle, seed, gapscale):
    valid, p3d, _ = _cloud(seed=seed)
    zgap, _ = oracle.zgap_percentile(valid, p3d, 90.0)
    zgap *= gapscale
    m = gpu_ctx.mesh_upload(valid, p3d)
    size = m.keep_biggest_component(zgap)
    v_ref, size_ref = oracle.keep_biggest_component(valid, p3d, zgap)
    v_got, _, _ = m.download()
    assert size == size_ref
    np.testing.assert_array_equal(v_got, v_ref)


def test_biggest_component_tie_break_and_empty(gpu_ctx, oracle):
    h, w = 30, 40
    valid = np.zeros((h, w), np.uint8); valid[20:25, 3:8] = 1; valid[2:7, 10:15] = 1
    p3d = np.zeros((h, w, 3))
    m = gpu_ctx.mesh_upload(valid, p3d)
    assert m.keep_biggest_component(1.0) == 25
    v_ref, _ = oracle.keep_biggest_component(valid, p3d, 1.0)
    np.testing.assert_array_equal(m.download()[0], v_ref)
    e = gpu_ctx.mesh_upload(np.zeros((5, 6), np.uint8), np.zeros((5, 6, 3)))
    assert e.keep_biggest_component(1.0) == 0 and not e.download()[0].any()


@pytest.mark.parametrize("central", [False, True])
def test_refinement_inliers_are_selected_on_the_device(gpu_ctx, central):
    """wass_mesh_refinement_inliers == every 10th point that passes refine_plane's tests (PovMesh.cpp:590-606), counted in
    raster order over the (optionally central-third) window -- what wass_stereo.cpp:2077-2085 writes to
    plane_refinement_inliers.xyz."""
    rng = np.random.default_rng(3)
    h, w = 123, 517                                               # several 256-point blocks per row, ragged last block
    p3d = rng.normal(0, 30, (h, w, 3))
    p3d[..., 2] += 60
    valid = (rng.random((h, w)) < 0.8).astype(np.uint8)
    m = gpu_ctx.mesh_upload(valid, p3d)
    kw = dict(xmin=-40.0, xmax=35.0, ymin=-50.0, ymax=45.0, max_distance=75.0)
    got = m.refinement_inliers(every=10, central_third_only=central, **kw)
    u0, u1, v0, v1 = (w // 4, w * 3 // 4, h // 4, h * 2 // 3) if central else (0, w - 1, 0, h - 1)
    uu, vv = np.meshgrid(np.arange(w), np.arange(h))
    x, y, z = p3d[..., 0], p3d[..., 1], p3d[..., 2]
    ok = (valid != 0) & (uu >= u0) & (uu <= u1) & (vv >= v0) & (vv <= v1) & (x > kw["xmin"]) & (x < kw["xmax"]) & \
         (y > kw["ymin"]) & (y < kw["ymax"]) & (np.sqrt(x * x + y * y + z * z) < kw["max_distance"])
    want = p3d[ok][::10]                                          # boolean indexing walks the grid in raster order
    np.testing.assert_array_equal(got, want)
    assert m.refinement_inliers(every=1, central_third_only=central, **kw).shape[0] == ok.sum()
    empty = gpu_ctx.mesh_upload(np.zeros((5, 6), np.uint8), np.zeros((5, 6, 3)))
    assert empty.refinement_inliers().shape == (0, 3)


def test_mesh_may_outlive_its_context():
    """A mesh destroyed after its context (garbage-collection order) frees its allocation instead of parking it for an owner
    that no longer exists; a new context then works normally."""
    import gc
    c2 = wass_amd.Context(0)
    m = c2.mesh_upload(np.ones((6, 7), np.uint8), np.zeros((6, 7, 3)))
    c2.close()
    del m
    gc.collect()
    c3 = wass_amd.Context(0)
    m3 = c3.mesh_upload(np.ones((6, 7), np.uint8), np.ones((6, 7, 3)))
    assert m3.download()[0].all()
    del m3
    c3.close()


def test_ransac_sampler_matches_oracle(oracle):
    a = wass_amd.ransac_sample(210, 140, 400, 12345)
    b = oracle.ransac_sample(210, 140, 400, 12345)
    np.testing.assert_array_equal(a, b)


def test_ransac_exact_counts_and_plane(gpu_ctx, oracle):
    valid, p3d, plane_true = _cloud(seed=4)
    uv = wass_amd.ransac_sample(valid.shape[1], valid.shape[0], 400, 12345)
    m = gpu_ctx.mesh_upload(valid, p3d)
    found, plane, best = m.ransac_plane(uv, 1.0)
    ok_ref, plane_ref, best_ref, per = oracle.ransac_plane(valid, p3d, uv, 1.0)
    assert found == ok_ref and best == best_ref          # same fp64 operations -> same inlier count
    np.testing.assert_array_equal(plane, plane_ref)
    assert abs(abs(plane[:3] @ plane_true[:3]) - 1) < 1e-2
    # failure path: best < W*H/10
    v2 = valid.copy(); v2[:] = 0; v2[:6, :6] = 1
    found2, _, best2 = gpu_ctx.mesh_upload(v2, p3d).ransac_plane(uv, 1.0)
    ok2, _, bref2, _ = oracle.ransac_plane(v2, p3d, uv, 1.0)
    assert found2 == ok2 == False and best2 == bref2   # noqa: E712


def test_crop_plane_exact(gpu_ctx, oracle):
    valid, p3d, plane = _cloud(seed=5)
    m = gpu_ctx.mesh_upload(valid, p3d)
    kept = m.crop_plane(plane, 0.2)
    v_ref, k_ref = oracle.crop_plane(valid, p3d, plane, 0.2)
    assert kept == k_ref
    np.testing.assert_array_equal(m.download()[0], v_ref)


@pytest.mark.parametrize("weighted,central", [(True, False), (False, False), (True, True)])
def test_refine_plane(gpu_ctx, oracle, weighted, central):
    valid, p3d, plane = _cloud(seed=6, outliers=0.0)
    m = gpu_ctx.mesh_upload(valid, p3d)
    got, n = m.refine_plane(xmin=-9.0, xmax=9999.0, weight_by_distance=weighted, central_third_only=central)
    ref, n_ref, _ = oracle.refine_plane(valid, p3d, xmin=-9.0, xmax=9999.0, weight_by_distance=weighted,
                                        central_third_only=central)
    assert n == n_ref
    # tolerance: fp64 sums in a different (tree) order; 1e-9 on a unit normal / d ~ 18
    np.testing.assert_allclose(got, ref, rtol=0, atol=1e-9)
    assert abs(got[:3] @ plane[:3]) > 0.999


def test_rt_from_plane_matches_oracle(oracle):
    pl = np.array([0.0123, -0.4567, 0.8895, -11.2]); pl[:3] /= np.linalg.norm(pl[:3])
    for a, b in zip(wass_amd.RT_from_plane(pl), oracle.RT_from_plane(pl)):
        np.testing.assert_array_equal(a, b)


def test_xyzc_bytes_exact(gpu_ctx, oracle):
    valid, p3d, plane = _cloud(seed=7)
    m = gpu_ctx.mesh_upload(valid, p3d)
    blob = m.encode_xyzc(plane)
    ref = oracle.encode_xyzc(valid, p3d, plane)
    assert len(blob) == len(ref) == 148 + 6 * int(valid.sum())
    assert blob == ref
    # format sanity (Appendix B.1) and the no-plane fallback
    n = struct.unpack("<I", blob[:4])[0]
    assert n == valid.sum()
    buf = np.empty(148 + 6 * valid.size, np.uint8)
    nb = m.encode_xyzc_to(plane, buf.ctypes.data, buf.size)
    assert buf[:nb].tobytes() == ref
    with pytest.raises(wass_amd.WassError):
        m.encode_xyzc_to(plane, buf.ctypes.data, 200)
    # asynchronous download into pinned memory: complete after synchronize(); back-to-back encodes reuse the
    # device-side payload buffer only after the previous download has finished
    import torch
    pins = [torch.zeros(148 + 6 * valid.size, dtype=torch.uint8).pin_memory() for _ in range(3)]
    sizes = [m.encode_xyzc_async(plane, p.data_ptr(), p.numel()) for p in pins]
    gpu_ctx.synchronize()
    for p, nb2 in zip(pins, sizes):
        assert p[:nb2].numpy().tobytes() == ref
    blob2 = m.encode_xyzc(None)
    Rinv = np.frombuffer(blob2[52:124], np.float64).reshape(3, 3)
    np.testing.assert_array_equal(Rinv, np.eye(3))


@pytest.mark.parametrize("seed", [0, 3])
def test_fused_calls_equal_step_by_step(gpu_ctx, oracle, seed):
    """wass_mesh_remove_outliers / wass_mesh_fit_plane take their decisions on the device; same results as the
    one-call-per-PovMesh-method sequence and as the oracle."""
    valid, p3d, _ = _cloud(seed=seed)
    uv = wass_amd.ransac_sample(valid.shape[1], valid.shape[0], 400, 12345)
    a = gpu_ctx.mesh_upload(valid, p3d)
    zg, ng = a.zgap_percentile(99.0)
    sz = a.keep_biggest_component(zg)
    found, pl, best = a.ransac_plane(uv, 1.0)
    k1 = a.crop_plane(pl, 1.0)
    pl2, ninl = a.refine_plane()
    k2 = a.crop_plane(pl2, 1.5)
    b = gpu_ctx.mesh_upload(valid, p3d)
    zg_b, ng_b, sz_b = b.remove_outliers(99.0)
    assert (zg_b, ng_b, sz_b) == (zg, ng, sz)
    res = b.fit_plane(uv, 1.0, 1.5)
    assert bool(res.found) == found and res.ransac_inliers == best
    np.testing.assert_array_equal(np.array(res.ransac_plane[:]), pl)
    assert res.kept_after_ransac_crop == k1 and res.refine_inliers == ninl and res.kept_final == k2
    np.testing.assert_array_equal(np.array(res.plane[:]), pl2)          # same arithmetic on host and device
    np.testing.assert_array_equal(a.download()[0], b.download()[0])
    # oracle
    ozg, _ = oracle.zgap_percentile(valid, p3d, 99.0)
    assert zg_b == ozg
    # RANSAC failure path: nothing cropped, plane NaN
    v2 = valid.copy(); v2[:] = 0; v2[:6, :6] = 1
    m2 = gpu_ctx.mesh_upload(v2, p3d)
    r2 = m2.fit_plane(uv, 1.0, 1.5)
    assert not r2.found and np.isnan(np.array(r2.plane[:])).all()
    np.testing.assert_array_equal(m2.download()[0], v2)
    # empty mesh
    e = gpu_ctx.mesh_upload(np.zeros((9, 11), np.uint8), np.zeros((9, 11, 3)))
    ze, ne, se = e.remove_outliers(99.0)
    assert np.isnan(ze) and ne == 0 and se == 0


@pytest.mark.parametrize("seed", [0, 3])
def test_async_frame_tail_equals_stage_by_stage(gpu_ctx, oracle, seed):
    """wass_mesh_finish_frame_async (no host synchronisation, every decision and the xyzC header on the device):
    same numbers and the same file bytes as the stage-by-stage calls; also the RANSAC-failure and empty cases."""
    import torch
    valid, p3d, _ = _cloud(seed=seed)
    h, w = valid.shape
    uv = wass_amd.ransac_sample(w, h, 400, 12345)
    a = gpu_ctx.mesh_upload(valid, p3d)
    zg, ng, sz = a.remove_outliers(99.0)
    res = a.fit_plane(uv, 1.0, 1.5)
    ref_bytes = a.encode_xyzc(np.array(res.plane[:]))
    pin = torch.zeros(148 + 6 * w * h, dtype=torch.uint8).pin_memory()
    for rep in range(2):                                     # twice: buffers and events are reused across frames
        b = gpu_ctx.mesh_upload(valid, p3d)
        b.finish_frame_async(uv, pin.data_ptr(), pin.numel())
        b.close()                                            # the mesh may be released right after the enqueue
        fr = gpu_ctx.frame_result()
        assert (fr.zgap, fr.n_gaps, fr.component_size) == (zg, ng, sz)
        assert fr.found == res.found and fr.refine_ok == 1 and fr.ransac_inliers == res.ransac_inliers
        np.testing.assert_array_equal(np.array(fr.ransac_plane[:]), np.array(res.ransac_plane[:]))
        np.testing.assert_array_equal(np.array(fr.plane[:]), np.array(res.plane[:]))
        assert (fr.kept_after_ransac_crop, fr.refine_inliers, fr.kept_final) == (res.kept_after_ransac_crop, res.refine_inliers, res.kept_final)
        assert fr.xyzc_bytes == len(ref_bytes) and fr.n_points == (len(ref_bytes) - 148) // 6
        assert pin[:fr.xyzc_bytes].numpy().tobytes() == ref_bytes
    # RANSAC failure: nothing cropped, NaN plane, identity R|T in the header -- the bytes of encode_xyzc(None)
    v2 = valid.copy(); v2[:] = 0; v2[:6, :6] = 1
    m2 = gpu_ctx.mesh_upload(v2, p3d)
    m2.remove_outliers(99.0)
    ref2 = m2.encode_xyzc(None)
    m3 = gpu_ctx.mesh_upload(v2, p3d)
    m3.finish_frame_async(uv, pin.data_ptr(), pin.numel())
    fr = gpu_ctx.frame_result()
    assert not fr.found and np.isnan(np.array(fr.plane[:])).all()
    assert pin[:fr.xyzc_bytes].numpy().tobytes() == ref2
    # empty mesh: zero points, the reference's +-DBL_MAX limits
    e = gpu_ctx.mesh_upload(np.zeros((9, 11), np.uint8), np.zeros((9, 11, 3)))
    ref3 = e.encode_xyzc(None)
    e.finish_frame_async(wass_amd.ransac_sample(11, 9, 8, 1), pin.data_ptr(), pin.numel())
    fr = gpu_ctx.frame_result()
    assert fr.n_points == 0 and pin[:148].numpy().tobytes() == ref3
    with pytest.raises(wass_amd.WassError):
        a.finish_frame_async(uv, pin.data_ptr(), 1000)         # needs room for every grid point


def test_planes_mean(oracle):
    import os
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "planes_txt.npz"))
    acc = wass_amd.planes_mean_accumulate(z["planes"][:2])
    acc = wass_amd.planes_mean_accumulate(z["planes"][2:], acc)
    mean, n = wass_amd.planes_mean_finish(acc)
    assert n == int(z["n_valid"])
    np.testing.assert_allclose(mean, z["nanmean"], rtol=1e-15)


def test_full_chain_small(gpu_ctx, oracle):
    """The whole path a1-a20 on one small synthetic frame: GPU chain vs oracle chain."""
    w, h, D = 320, 240, 64
    right, left = synth.make_pair(w, h, D, frame_idx=21)
    p = default_sgm_params(D, ndirs=5)
    rig = synth.rig_geometry(w, h)
    roi = (0, 0, w, h)
    # GPU
    d16 = gpu_ctx.sgm_disparity(right, left, p)
    f = gpu_ctx.disparity_postprocess(d16, p)
    mesh, n = gpu_ctx.triangulate(f, w, h, roi, roi, wass_amd.make_geom(rig), right, None, (right <= 254).astype(np.uint8))
    zg, _ = mesh.zgap_percentile(99.0)
    mesh.keep_biggest_component(zg)
    uv = wass_amd.ransac_sample(w, h, 400, 12345)
    found, pl, best = mesh.ransac_plane(uv, 1.0)
    assert found
    mesh.crop_plane(pl, 1.0)
    pl2, ninl = mesh.refine_plane()
    mesh.crop_plane(pl2, 1.5)
    blob = mesh.encode_xyzc(pl2)
    # oracle
    od16, _ = oracle.dense_disparity16(right, left, _oracle_params(oracle, p))
    of = oracle.disparity_postprocess(od16, 1, D)
    on, ov, op3, og = oracle.triangulate(of, roi, roi, oracle.make_geom(rig), right, None, (right <= 254).astype(np.uint8))
    ozg, _ = oracle.zgap_percentile(ov, op3, 99.0)
    ov, _ = oracle.keep_biggest_component(ov, op3, ozg)
    ok, opl, obest, _ = oracle.ransac_plane(ov, op3, uv, 1.0)
    ov, _ = oracle.crop_plane(ov, op3, opl, 1.0)
    opl2, oninl, _ = oracle.refine_plane(ov, op3)
    ov, _ = oracle.crop_plane(ov, op3, opl2, 1.5)
    np.testing.assert_array_equal(d16, od16)
    np.testing.assert_array_equal(f, of)
    assert n == on and zg == ozg and best == obest and ninl == oninl
    np.testing.assert_array_equal(pl, opl)
    # refined plane: 1e-9 absolute (unit normal, |d| ~ 20 baseline units) = far below the xyzC quantisation
    np.testing.assert_allclose(pl2, opl2, rtol=0, atol=1e-9)
    v_got = mesh.download()[0]
    assert (v_got != ov).sum() <= 2                      # a point within 1e-9 of the 1.5 threshold may flip
    # the plane must be the synthetic sea plane: affine disparity <=> planar surface
    assert n > 0.7 * w * h and best > 0.8 * n
    assert len(blob) == 148 + 6 * int(v_got.sum())


def test_pipelined_frames_equal_serial_frames(oracle):
    """wass_ctx_set_tail_overlap: the post-SGM stages of frame i run on the tail stream underneath the SGM stage of
    frame i+1.  Same planes and the same mesh_cam.xyzC bytes as the serial order, with one disparity buffer reused
    for every frame (ordering enforced by the library) and with two alternating ones."""
    import torch
    w, h, D = 320, 240, 64
    dev = torch.device("cuda", 0)
    p = default_sgm_params(D, ndirs=8)
    geom = wass_amd.make_geom(synth.rig_geometry(w, h))
    roi = (0, 0, w, h)
    uv = wass_amd.ransac_sample(w, h, 400, 12345)
    frames = [tuple(torch.from_numpy(a).to(dev) for a in synth.make_pair(w, h, D, frame_idx=k)) for k in range(5)]

    def run(overlap, nbuf):
        ctx = wass_amd.Context(0)
        ctx.set_tail_overlap(overlap)
        outs = [torch.empty((h, w), dtype=torch.int16, device=dev) for _ in range(nbuf)]
        dispf = torch.empty((h, w), dtype=torch.float32, device=dev)
        pins = [torch.zeros(148 + 6 * w * h, dtype=torch.uint8).pin_memory() for _ in range(2)]
        res = []
        for i, (dr, dl) in enumerate(frames):
            out = outs[i % nbuf]
            ctx.sgm_disparity_dev(dr, dl, p, out)
            ctx.disparity_postprocess_dev(out, p, 1, 2, 0, dispf)
            mesh, _ = ctx.triangulate_dev(dispf, w, h, roi, roi, geom, dr, None, None, 20.0, None, 1.0, count=False)
            if i > 0:
                fr = ctx.frame_result()
                res.append((np.array(fr.plane[:]), pins[(i - 1) % 2][:fr.xyzc_bytes].numpy().tobytes()))
            mesh.finish_frame_async(uv, pins[i % 2].data_ptr(), pins[i % 2].numel())
            mesh.close()
        fr = ctx.frame_result()
        res.append((np.array(fr.plane[:]), pins[(len(frames) - 1) % 2][:fr.xyzc_bytes].numpy().tobytes()))
        ctx.close()
        return res

    serial = run(False, 1)
    assert len({b for _, b in serial}) == len(frames)          # five different frames, five different files
    for variant in (run(True, 1), run(True, 2)):
        for (pa, ba), (pb, bb) in zip(serial, variant):
            np.testing.assert_array_equal(pa, pb)
            assert ba == bb


def test_frame_pipeline_matches_stage_by_stage(gpu_ctx, oracle):
    """wass_amd.batch.FramePipeline (what bench.py and a sequence driver run): every frame's plane and file bytes equal
    the stage-by-stage chain on a second context, outputs arrive one frame late and in order."""
    import torch
    from wass_amd.batch import FramePipeline
    w, h, D = 320, 240, 64
    dev = torch.device("cuda", 0)
    p = default_sgm_params(D, ndirs=5)
    geom = wass_amd.make_geom(synth.rig_geometry(w, h))
    roi = (0, 0, w, h)
    frames = [tuple(torch.from_numpy(a).to(dev) for a in synth.make_pair(w, h, D, frame_idx=30 + k)) for k in range(4)]
    masks = [(fr[0] <= 254).to(torch.uint8) for fr in frames]
    uv = wass_amd.ransac_sample(w, h, 400, 12345)
    ref = []
    for (dr, dl), m in zip(frames, masks):
        d16 = gpu_ctx.sgm_disparity_dev(dr, dl, p)
        f = gpu_ctx.disparity_postprocess_dev(d16, p, 1, 2, 0)
        mesh, _ = gpu_ctx.triangulate_dev(f, w, h, roi, roi, geom, dr, None, m, 20.0, None, 1.0)
        mesh.remove_outliers(99.0)
        res = mesh.fit_plane(uv, 1.0, 1.5)
        ref.append((np.array(res.plane[:]), mesh.encode_xyzc(np.array(res.plane[:]))))
    with wass_amd.Context(0) as ctx2:
        pipe = FramePipeline(ctx2, w, h, p, geom)
        outs = []
        for (dr, dl), m in zip(frames, masks):
            o = pipe.submit(dr, dl, d_right_mask=m)
            if o is not None:
                outs.append((o.index, o.plane.copy(), o.xyzc.tobytes()))
        for o in pipe.drain():
            outs.append((o.index, o.plane.copy(), o.xyzc.tobytes()))
        assert pipe.flush() is None and pipe.drain() == []
    assert [i for i, _, _ in outs] == [0, 1, 2, 3]
    for (_, pl, by), (rpl, rby) in zip(outs, ref):
        np.testing.assert_array_equal(pl, rpl)
        assert by == rby


@pytest.mark.gpu
def test_inlier_text_from_the_device_equals_printf(gpu_ctx):
    """wass_mesh_finish_frame_async_ex2: the text of plane_refinement_inliers.xyz as the device formats it (csrc/fmt_g6.h) is, byte for
    byte, what printf("%g %g %g\\n") makes of the selected points -- the reference's default-ofstream output (wass_stereo.cpp:2077-2085)."""
    import torch
    import wass_amd
    from wass_amd import synth
    from wass_amd.batch import FramePipeline
    w, h, D = 640, 480, 64
    dev = torch.device("cuda", 0)
    params = wass_amd.default_sgm_params(D, ndirs=5)
    geom = wass_amd.make_geom(synth.rig_geometry(w, h))
    pipe = FramePipeline(gpu_ctx, w, h, params, geom, inliers_text=True, keep_inlier_points=True)
    outs = []
    for k in range(3):
        r, l = synth.make_pair_torch(w, h, D, frame_idx=40 + k, device=dev)
        mask = (r <= 254).to(torch.uint8)
        o = pipe.submit(r, l, d_right_image=r, d_right_mask=mask)
        if o is not None:
            outs.append((o.inliers_xyz.copy(), bytes(o.inliers_text)))
    for o in pipe.drain():
        outs.append((o.inliers_xyz.copy(), bytes(o.inliers_text)))
    assert len(outs) == 3
    for xyz, text in outs:
        assert len(xyz) > 1000
        assert text == "".join("%g %g %g\n" % tuple(p) for p in xyz).encode()


def test_two_frames_may_be_pending_and_records_come_back_in_order(gpu_ctx):
    """wass_mesh_finish_frame_async*: two frames may be pending per context (a driver enqueues frame n+1's tail before it reads frame
    n's record); a third call without a read in between is refused; wass_ctx_frame_result hands the records out in submission order
    and refuses when nothing is pending."""
    import torch
    clouds = [_cloud(seed=k) for k in (0, 3, 5)]
    h, w = clouds[0][0].shape
    uv = wass_amd.ransac_sample(w, h, 400, 12345)
    ref = []
    for valid, p3d, _ in clouds:                               # each frame alone, stage by stage
        a = gpu_ctx.mesh_upload(valid, p3d)
        a.remove_outliers(99.0)
        res = a.fit_plane(uv, 1.0, 1.5)
        ref.append(a.encode_xyzc(np.array(res.plane[:])))
    assert len({len(b) for b in ref}) == 3                     # three different frames
    pins = [torch.zeros(148 + 6 * w * h, dtype=torch.uint8).pin_memory() for _ in range(3)]
    m = [gpu_ctx.mesh_upload(v, p) for v, p, _ in clouds]
    m[0].finish_frame_async(uv, pins[0].data_ptr(), pins[0].numel()); m[0].close()
    m[1].finish_frame_async(uv, pins[1].data_ptr(), pins[1].numel()); m[1].close()
    with pytest.raises(wass_amd.WassError):                    # a third pending frame is refused ...
        m[2].finish_frame_async(uv, pins[2].data_ptr(), pins[2].numel())
    r0 = gpu_ctx.frame_result()                                # ... until the oldest record has been read
    m[2].finish_frame_async(uv, pins[2].data_ptr(), pins[2].numel()); m[2].close()
    r1 = gpu_ctx.frame_result()
    r2 = gpu_ctx.frame_result()
    with pytest.raises(wass_amd.WassError):
        gpu_ctx.frame_result()                                 # nothing pending any more
    for k, r in enumerate((r0, r1, r2)):
        assert r.found and r.xyzc_bytes == len(ref[k])
        assert pins[k][:r.xyzc_bytes].numpy().tobytes() == ref[k]


def test_sgm_call_timings_serve_the_last_four_calls(gpu_ctx):
    """wass_sgm_call_timings: a driver two frames ahead reads the event times of call n-2 while n is running; four sets are kept."""
    rng = np.random.default_rng(5)
    r = rng.integers(0, 256, (96, 256), dtype=np.uint8)
    l = np.roll(r, 7, axis=1)
    p = default_sgm_params(64, ndirs=8)
    for _ in range(6):
        gpu_ctx.sgm_disparity(r, l, p)
    n = gpu_ctx.sgm_call_count()
    assert n >= 6
    for call in (n, n - 1, n - 2, n - 3):
        assert gpu_ctx.sgm_call_timings(call).total_ms > 0
    for call in (n + 1, n - 4, 0):
        with pytest.raises(wass_amd.WassError):
            gpu_ctx.sgm_call_timings(call)
