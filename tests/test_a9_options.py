"""Row a9 options of sgbm_dense_stereo: DENSE_SCALE != 1 (wass_stereo.cpp:788-796, 853, 903-904), the biggest component by
disparity gradient (:947-986) and cv::filterSpeckles (DENSE_SPECKLE_WINDOW_SIZE > 0).  CPU: known answers for the oracle's
restatement of the OpenCV functions involved (oracle/a9_oracle.c; parity unpinned).  GPU: bit-exact against that oracle."""
import os
import subprocess

import numpy as np
import pytest

from wass_amd import default_sgm_params, synth


# ------------------------------------------------------------------ CPU: the oracle's restatement
def test_resize_destination_size_rounds_like_cvround(oracle):
    assert oracle.resize_dsize(2456, 2058, 0.5, 0.5) == (1228, 1029)
    assert oracle.resize_dsize(101, 51, 0.5, 0.5) == (50, 26)           # 50.5 -> 50 (half to even), 25.5 -> 26
    assert oracle.resize_dsize(100, 40, 1.3, 1.0) == (130, 40)


def test_cubic_resize_keeps_constants_and_ramps(oracle):
    flat = np.full((31, 47), 200, np.uint8)
    for f in (0.5, 0.75, 1.5):
        assert (oracle.resize_cubic_u8(flat, f, f) == 200).all()
    ramp = np.tile(np.arange(20, 220, 2, dtype=np.uint8), (16, 1))       # I(x) = 20 + 2x: a cubic kernel reproduces it
    half = oracle.resize_cubic_u8(ramp, 0.5, 0.5)
    x = np.arange(half.shape[1])
    expect = 20 + 2 * ((x + 0.5) * 2 - 0.5)                              # source coordinate of destination pixel x
    assert np.abs(half[4, 2:-2].astype(float) - expect[2:-2]).max() <= 1.0
    wide = oracle.resize_cubic_u8(ramp, 2.0, 1.0)                        # DENSE_SCALE > 1 stretches x only
    assert wide.shape == (16, 200) and (wide[3] == wide[9]).all()


def test_nearest_and_speckles_and_components_known_answers(oracle):
    # two plateaus joined by a steep edge, a small island, a large island
    d = np.zeros((20, 30), np.float32)
    d[2:12, 2:14] = 10.0
    d[2:12, 14:20] = 40.0                   # the jump 10 -> 40 has a squared Sobel gradient far above the threshold
    d[15:18, 3:6] = 12.0                    # 9-pixel island
    out, area = oracle.biggest_component_by_gradient(d, 400)
    assert (out[15:18, 3:6] == 0).all() and (out[3:11, 3:12] == 10.0).all() and (out[:, 13:15] == 0).all()
    assert area == int((out != 0).sum()) and 0 < area < 120
    # speckles: a 2x2 blob and a 1-pixel outlier vanish (size <= 4), the 5x5 region stays, the invalid value is never a region
    s = np.full((12, 12), -16, np.int16)
    s[1:6, 1:6] = 160
    s[1, 1] = 170                           # within maxDiff of its neighbours: same region
    s[8:10, 8:10] = 320
    s[10, 2] = 500
    f = oracle.filter_speckles(s, -16, 4, 16)
    assert (f[1:6, 1:6] == s[1:6, 1:6]).all() and (f[8:10, 8:10] == -16).all() and f[10, 2] == -16
    s2 = s.copy(); s2[3, 3] = 400           # an outlier INSIDE the big region is its own 1-pixel region
    f2 = oracle.filter_speckles(s2, -16, 4, 16)
    assert f2[3, 3] == -16 and f2[2, 2] == 160


def test_postprocess_ex_reduces_to_the_plain_form(oracle):
    right, left = synth.make_pair(96, 64, 32, frame_idx=4)
    d16, _ = oracle.dense_disparity16(right, left, oracle.wass_params(32))
    np.testing.assert_array_equal(oracle.disparity_postprocess_ex(d16, 1, 32, 96, 64), oracle.disparity_postprocess(d16, 1, 32))


# ------------------------------------------------------------------ GPU parity
def _oracle_params(O, p):
    return O.SgbmParams(p.min_disp, p.num_disp, p.win, p.P1, p.P2, p.uniq_ratio, p.disp12_max_diff,
                        p.prefilter_cap, p.speckle_win, p.speckle_range, p.ndirs)


@pytest.mark.gpu
@pytest.mark.timeout(180)
@pytest.mark.parametrize("scale", [0.5, 0.73, 2.0, 1.37])
def test_dense_scale_sgm_and_cleanup(gpu_ctx, oracle, scale):
    w, h, D = 211, 133, 32
    right, left = synth.make_pair(w, h, D, frame_idx=17)
    p = default_sgm_params(D, ndirs=5)
    p.dense_scale = scale
    got = gpu_ctx.sgm_disparity(right, left, p)
    r2, l2 = oracle.dense_inputs(right, left, scale)
    assert got.shape == r2.shape
    ref, st = oracle.dense_disparity16(r2, l2, _oracle_params(oracle, p))
    assert not st.overflow
    np.testing.assert_array_equal(got, ref)
    # the device-pointer entry allocates / checks its output at the size of the RESIZED inputs (a (h, w) tensor would be
    # overrun by the median / crop kernel for scale > 1)
    import torch
    dev = torch.device("cuda", gpu_ctx.device_id)
    d = gpu_ctx.sgm_disparity_dev(torch.from_numpy(right).to(dev), torch.from_numpy(left).to(dev), p)
    gpu_ctx.synchronize()
    assert tuple(d.shape) == r2.shape
    np.testing.assert_array_equal(d.cpu().numpy(), ref)
    with pytest.raises(ValueError):
        gpu_ctx.sgm_disparity_dev(torch.from_numpy(right).to(dev), torch.from_numpy(left).to(dev), p,
                                  torch.empty((h, w), dtype=torch.int16, device=dev))
    for cc in (0, 30):
        f = gpu_ctx.disparity_postprocess_ex(got, p, w, h, cc_threshold=cc)
        np.testing.assert_array_equal(f, oracle.disparity_postprocess_ex(ref, 1, D, w, h, dense_scale=scale, cc_threshold=cc))
    assert (f != 0).mean() > 0.3


@pytest.mark.gpu
@pytest.mark.timeout(180)
@pytest.mark.parametrize("ndirs,off", [(5, 0), (8, 0), (5, 3)])
def test_speckle_filter_inside_sgbm(gpu_ctx, oracle, ndirs, off):
    rng = np.random.default_rng(3)
    w, h, D = 180, 90, 32
    right, left = synth.make_pair(w, h, D, frame_idx=2)
    right = right.copy(); right[rng.random((h, w)) < 0.08] = 255        # salt noise: isolated wrong disparities
    p = default_sgm_params(D, ndirs=ndirs, disp_offset=off, win=3, p2_mult=8)     # weak smoothing: the noise survives as speckles
    p.speckle_win, p.speckle_range = 60, 2
    got = gpu_ctx.sgm_disparity(right, left, p)
    ref, st = oracle.dense_disparity16(right, left, _oracle_params(oracle, p), off)
    np.testing.assert_array_equal(got, ref)
    p.speckle_win = -70
    assert (gpu_ctx.sgm_disparity(right, left, p) != got).any(), "the filter must have removed something"


@pytest.mark.gpu
@pytest.mark.timeout(180)
def test_biggest_component_by_gradient_on_random_blobs(gpu_ctx, oracle):
    import torch
    rng = np.random.default_rng(9)
    for (w, h) in ((97, 61), (256, 40)):
        d = np.where(rng.random((h, w)) < 0.62, rng.uniform(5, 6, (h, w)), 0.0).astype(np.float32)
        d[h // 3:h // 3 + 6, :] += 20.0                                  # a ridge with a large gradient
        for thr in (1, 50):
            t = torch.from_numpy(d.copy()).cuda()
            rc = gpu_ctx._lib.wass_biggest_component_by_gradient_dev(gpu_ctx._h, t.data_ptr(), w, h, thr)
            assert rc == 0
            gpu_ctx.synchronize()
            np.testing.assert_array_equal(t.cpu().numpy(), oracle.biggest_component_by_gradient(d, thr)[0])


@pytest.mark.gpu
@pytest.mark.timeout(180)
def test_cli_accepts_the_three_options(tmp_path):
    """DENSE_SCALE, DENSE_DISPARITY_BIGGEST_COMPONENT_THRESHOLD and DENSE_SPECKLE_WINDOW_SIZE used to be hard errors."""
    from test_cli import make_workdir
    from wass_amd import build
    cli = build.build_host()
    wd, cfg, *_ = make_workdir(str(tmp_path), 320, 240, 64, extra_cfg="DENSE_SCALE=0.5\nDENSE_DISPARITY_BIGGEST_COMPONENT_THRESHOLD=900\n"
                                                                        "DENSE_SPECKLE_WINDOW_SIZE=40\n")
    r = subprocess.run([cli, cfg, wd], capture_output=True, text=True, env=dict(os.environ, WASS_DEBUG_FORMAT="png"))
    assert r.returncode == 0, r.stdout
    assert "Dense-stereo input resize: [320 x 240] -> [160 x 120]" in r.stdout
    # the option's two debug pictures (wass_stereo.cpp:958-960, 981-983): 255 on large gradients / outside the kept component
    from test_cli import _read_png
    lg = _read_png(os.path.join(wd, "disparity_large_gradient.png")); nb = _read_png(os.path.join(wd, "disparity_biggest_component.png"))
    fs = _read_png(os.path.join(wd, "disparity_final_scaled.png"))
    assert lg.shape == nb.shape == fs.shape and set(np.unique(lg)) <= {0, 255} and set(np.unique(nb)) <= {0, 255}
    assert 0 < (lg == 255).mean() < 0.5 and 0.05 < (nb == 0).mean() < 0.99
    assert (nb[lg == 255] == 255).all()                            # a large-gradient pixel is zeroed, so it is never in the component
    assert "extracting the biggest connected component" in r.stdout
    n = int.from_bytes(open(os.path.join(wd, "mesh_cam.xyzC"), "rb").read(4), "little")
    assert n > 20000
    plane = [float(x) for x in open(os.path.join(wd, "plane.txt")).read().split()]
    assert abs(plane[1] - 0.8198) < 0.06 and abs(plane[2] - 0.5726) < 0.06            # the synthetic sea plane, at half resolution
