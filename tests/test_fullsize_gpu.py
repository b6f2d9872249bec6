"""BASELINE.json full-size configurations on the GPU: size-independent properties plus one full-size oracle run.

* config B (2456x2058, D=256), 5-path: bit-exact against the CPU oracle (about 20 s of CPU).
* config B and config E (3840x2160, D=512), 8-path: exact vertical-flip equivariance.  The 8-path set, the
  replicate-clamped cost windows, the per-pixel selection, the per-row L-R check and the 3x3 median are all closed
  under a vertical flip, and every tie is broken by d or x, never by y -- so flipping both inputs must flip the
  output bit for bit.  (The 5-path set is not closed: it only has top-down diagonals.)
* determinism (same input twice), accuracy against the synthetic ground truth, no int16 cost overflow.
"""
import numpy as np
import pytest

from wass_amd import default_sgm_params, synth

pytestmark = pytest.mark.gpu


def _oracle_params(O, p):
    return O.SgbmParams(p.min_disp, p.num_disp, p.win, p.P1, p.P2, p.uniq_ratio, p.disp12_max_diff,
                        p.prefilter_cap, p.speckle_win, p.speckle_range, p.ndirs)


@pytest.fixture(scope="module")
def pair_b():
    return synth.make_pair(2456, 2058, 256, frame_idx=0)


def test_config_b_5path_bit_exact_vs_oracle(gpu_ctx, oracle, pair_b):
    right, left = pair_b
    p = default_sgm_params(256, ndirs=5)
    got = gpu_ctx.sgm_disparity(right, left, p)
    ref, st = oracle.dense_disparity16(right, left, _oracle_params(oracle, p))
    assert not st.overflow
    np.testing.assert_array_equal(got, ref)
    np.testing.assert_array_equal(gpu_ctx.disparity_postprocess(got, p), oracle.disparity_postprocess(ref, 1, 256))


def test_config_b_8path_bit_exact_vs_oracle(gpu_ctx, oracle, pair_b):
    """The benchmark workload itself (2456x2058, D=256, 8 paths = MODE_HH): every disparity equals the oracle's."""
    right, left = pair_b
    p = default_sgm_params(256, ndirs=8)
    got = gpu_ctx.sgm_disparity(right, left, p)
    ref, st = oracle.dense_disparity16(right, left, _oracle_params(oracle, p))
    assert not st.overflow
    np.testing.assert_array_equal(got, ref)


def _flip_property(ctx, right, left, D):
    p = default_sgm_params(D, ndirs=8)
    a = ctx.sgm_disparity(right, left, p)
    b = ctx.sgm_disparity(np.ascontiguousarray(right[::-1]), np.ascontiguousarray(left[::-1]), p)
    np.testing.assert_array_equal(a, b[::-1])
    np.testing.assert_array_equal(a, ctx.sgm_disparity(right, left, p))          # deterministic
    assert ctx.sgm_timings().cost_overflow == 0
    return a


def test_config_b_8path_flip_equivariance_and_accuracy(gpu_ctx, pair_b):
    right, left = pair_b
    a = _flip_property(gpu_ctx, right, left, 256)
    gt = synth.true_disparity(2456, 2058, 256)
    f = a.astype(np.float32) / 16.0
    m = a > 16
    assert m.mean() > 0.9
    assert np.abs(f[m] - gt[m]).mean() < 0.25


def test_config_e_4k_512_disparities(gpu_ctx):
    w, h, D = 3840, 2160, 512
    right, left = synth.make_pair(w, h, D, frame_idx=5)
    a = _flip_property(gpu_ctx, right, left, D)
    gt = synth.true_disparity(w, h, D)
    m = a > 16
    assert m.mean() > 0.85
    assert np.abs(a[m] / 16.0 - gt[m]).mean() < 0.3


def test_wass_default_640_disparities_band(gpu_ctx, oracle):
    """MAX_DISPARITY=640 (the WASS default, NP=5) on a band the oracle finishes quickly."""
    w, h, D = 1400, 96, 640
    right, left = synth.make_pair(w, h, D, frame_idx=9)
    for nd in (5, 8):
        p = default_sgm_params(D, ndirs=nd)
        got = gpu_ctx.sgm_disparity(right, left, p)
        ref, st = oracle.dense_disparity16(right, left, _oracle_params(oracle, p))
        assert not st.overflow
        np.testing.assert_array_equal(got, ref)
