"""De-risking the unpinned SGBM oracle.

OpenCV is not available here, so oracle/sgbm_oracle.c (the oracle of record, against which every GPU kernel is
checked) cannot be pinned by the reference itself.  What this file adds:
  1. oracle/sgbm_direct.py, an independent direct-form numpy restatement written from SURVEY.md Appendix A only,
     must agree with sgbm_oracle.c on C, S, the raw disparity and the final map;
  2. hand-computed literal vectors (worked out on paper in the comments) for one Birchfield-Tomasi cell, one path step
     with sentinels, one sub-pixel division with a negative numerator and one left-right rejection;
  3. one test per item of the uncertainty register (Appendix F), named after the assumption it encodes -- if real
     OpenCV ever disagrees, the failing assumption is named by the test that has to change.
"""
import numpy as np
import pytest

from oracle import sgbm_direct as SD
from wass_amd import synth


def _pad(right, left, D):
    h, w = right.shape
    R = np.zeros((h, w + D), np.uint8); L = np.zeros((h, w + D), np.uint8)
    R[:, D:] = right; L[:, D:] = left
    return R, L


def _direct(img1, img2, p, mode):
    return SD.compute(img1, img2, p.min_disp, p.num_disp, p.block_size, p.P1, p.P2, p.uniqueness_ratio,
                      p.disp12_max_diff, p.prefilter_cap, mode)


# ------------------------------------------------------------------ 1. two independent restatements agree
@pytest.mark.parametrize("w,h,D,mode,win,mind", [
    (64, 48, 16, 5, 13, 1), (64, 48, 16, 8, 13, 1),           # the small committed fixture size
    (160, 120, 32, 5, 13, 1), (160, 120, 32, 8, 13, 1),
    (50, 30, 16, 8, 5, 0), (47, 21, 32, 5, 3, 2), (40, 12, 16, 8, 1, 1),
])
def test_direct_form_restatement_agrees_with_the_c_oracle(oracle, w, h, D, mode, win, mind):
    right, left = synth.make_pair(w, h, D, frame_idx=w + h)
    R, L = _pad(right, left, D)
    p = oracle.wass_params(D, mode=mode, win=win, min_disp=mind)
    disp, st, Co, So, rawo = oracle.sgbm_compute(R, L, p, dump=True)
    assert not st.overflow
    d2, C2, S2, raw2 = _direct(R, L, p, mode)
    np.testing.assert_array_equal(Co, C2, err_msg="cost volume C (A.2-A.3)")
    np.testing.assert_array_equal(So, S2, err_msg="aggregated volume S (A.4)")
    np.testing.assert_array_equal(rawo, raw2, err_msg="selection / uniqueness / sub-pixel / L-R (A.5)")
    np.testing.assert_array_equal(disp, d2, err_msg="median (A.6)")


@pytest.mark.parametrize("mode", [5, 8])
def test_restatements_agree_on_noise_with_ties_and_rejections(oracle, mode):
    """Random images: many ties in the minimum, failed uniqueness tests and L-R rejections."""
    rng = np.random.default_rng(11)
    D = 16
    R = rng.integers(0, 256, (24, 70), dtype=np.uint8); L = rng.integers(0, 256, (24, 70), dtype=np.uint8)
    p = oracle.wass_params(D, mode=mode, win=3, p2_mult=16)
    disp, st, Co, So, rawo = oracle.sgbm_compute(R, L, p, dump=True)
    assert not st.overflow
    d2, C2, S2, raw2 = _direct(R, L, p, mode)
    for a, b in ((Co, C2), (So, S2), (rawo, raw2), (disp, d2)):
        np.testing.assert_array_equal(a, b)
    inv = (p.min_disp - 1) * 16
    assert (rawo == inv).mean() > 0.05 and (rawo != inv).mean() > 0.05        # both branches are exercised


# ------------------------------------------------------------------ 2. literal vectors
def _bt_images():
    # every row identical, so the x-Sobel is 4 * (I[X+1] - I[X-1]); minD = 0, D = 16 -> maxD = 16, w = 20 -> width1 = 4
    i1 = np.full(20, 100, np.int64); i1[17] = 110; i1[18] = 130
    i2 = np.full(20, 100, np.int64); i2[15] = 120
    return np.tile(i1, (3, 1)).astype(np.uint8), np.tile(i2, (3, 1)).astype(np.uint8)


def test_literal_birchfield_tomasi_cell(oracle):
    """x = 1 (X = 17), d = 2 (X2 = 15), ftzero = 61:
    Sobel channel img1: X=16 -> 4*10 = 40 -> 101;  X=17 -> 4*30 = 120 -> clip 61 -> 122;  X=18 -> 4*(100-110) -> 21
       u = 122, half-pixel values (122+101)/2 = 111 and (122+21)/2 = 71  ->  [u0,u1] = [71,122]
    Sobel channel img2: X=14 -> 4*20 = 80 -> clip -> 122;  X=15 -> 0 -> 61;  X=16 -> -80 -> clip -> 0
       v = 61, half-pixel values (61+122)/2 = 91 and (61+0)/2 = 30     ->  [v0,v1] = [30,91]
       c0 = max(0, u - v1, v0 - u) = max(0, 31, -92) = 31;  c1 = max(0, v - u1, u0 - v) = max(0, -61, 10) = 10 -> 10
    raw channel: u = 110, [105,120];  v = 120, [110,120]: c0 = max(0, 110-120, 110-110) = 0 -> 0 >> 2 = 0
    pix = 10 + 0 = 10."""
    img1, img2 = _bt_images()
    p = SD.derived(0, 16, 1, 8, 32, 1, -1, 60)
    pix = SD.pixel_cost(img1, img2, p)
    assert pix.shape == (3, 4, 16) and (pix[:, 1, 2] == 10).all()
    # the C oracle with a 1x1 window: C = pix
    op = oracle.SgbmParams(0, 16, 1, 8, 32, 1, -1, 60, -70, 16, 5)
    _, _, Co, _, _ = oracle.sgbm_compute(img1, img2, op, dump=True)
    assert (Co[:, 1, 2] == 10).all()
    np.testing.assert_array_equal(Co, pix)


def test_literal_path_step_with_sentinels():
    """P1 = 3, P2 = 7.  Pixel 0 has no predecessor (all-zero state, min 0): L = C(0) + min(0, 0+3, 0+7) = C(0) = [6,4,9,5].
    Pixel 1, C = [1,1,1,1], predecessor [6,4,9,5], min 4, delta' = 4 + 7 = 11:
      d=0: min(6, MAX+3, 4+3, 11) = 6 -> 1 + 6 - 4 = 3      d=1: min(4, 6+3, 9+3, 11) = 4 -> 1
      d=2: min(9, 4+3, 5+3, 11) = 7 -> 4                    d=3: min(5, 9+3, MAX+3, 11) = 5 -> 2"""
    p = SD.derived(0, 4, 1, 3, 7, 1, -1, 60)
    C = np.array([[[6, 4, 9, 5], [1, 1, 1, 1]]], np.int64)
    L = SD.path_costs(C, p, (-1, 0))
    assert L[0, 0].tolist() == [6, 4, 9, 5] and L[0, 1].tolist() == [3, 1, 4, 2]
    # one-row image, MODE_SGBM: the three paths from the row above see the all-zero state (L = C); the right-to-left path
    # starts at pixel 1 (L = C(1)) and reaches pixel 0 with predecessor [1,1,1,1] (min 1): C(0) + min(1, 1+3, 8) - 1 = C(0)
    S, maxL = SD.aggregate(C, p, 5)
    assert S[0, 1].tolist() == [3 + 4, 1 + 4, 4 + 4, 2 + 4] and S[0, 0].tolist() == [30, 20, 45, 25] and maxL == 9


def test_literal_subpixel_division_truncates_toward_zero():
    """S = [100, 90, 130, 500], best = 1: denom2 = 100 + 130 - 180 = 50; numerator (100-130)*16 + 50 = -430;
    -430 / 100 = -4 in C (a flooring division would give -5); d16 = 1*16 - 4 = 12."""
    p = SD.derived(0, 4, 1, 3, 7, 1, -1, 60)
    d1 = SD.select_row(np.array([[100, 90, 130, 500]], np.int64), 5, p)
    assert d1.tolist() == [-16, -16, -16, -16, 12]


def test_literal_left_right_rejection():
    """D = 8, minD = 0, three pixels (X = 8, 9, 10).  x=2 picks d=2 (cost 100) and x=0 picks d=0 (cost 50): both point at
    right-view column 8, the cheaper one (d=0) keeps it.  The check at X=10 then finds disp2[8] = 0, |0 - 2| > 1 on both
    the floor and the ceil position (they coincide: d16 = 32 exactly) -> rejected.  x=1 (d=5 -> column 4) is consistent."""
    p = SD.derived(0, 8, 1, 3, 7, 1, -1, 60)
    S = np.array([[50, 900, 900, 900, 900, 900, 900, 900],
                  [900, 900, 900, 900, 900, 80, 900, 900],
                  [900, 500, 100, 500, 900, 900, 900, 900]], np.int64)
    d1 = SD.select_row(S, 11, p)
    assert d1.tolist() == [-16] * 8 + [0, 80, -16]


# ------------------------------------------------------------------ 3. Appendix F, one test per assumption
def _textured(oracle, w=90, h=40, D=16):
    right, left = synth.make_pair(w, h, D, frame_idx=5)
    left = left.copy(); left[10:25, 30:50] = np.roll(left[10:25, 30:50], 7, axis=1)     # a patch that breaks L-R consistency
    return _pad(right, left, D)


def test_assumption_disp12maxdiff_not_positive_means_tolerance_one_and_never_disables_the_check(oracle):
    R, L = _textured(oracle)
    outs = {}
    for d12 in (-1, 0, 1, 1000):
        p = oracle.wass_params(16)
        p.disp12_max_diff = d12
        outs[d12] = oracle.sgbm_compute(R, L, p, dump=True)[4]
    np.testing.assert_array_equal(outs[-1], outs[1])
    np.testing.assert_array_equal(outs[0], outs[1])
    assert (outs[1000] != outs[-1]).any(), "with an unreachable tolerance some rejected pixels must come back"


def test_assumption_border_columns_of_both_prefiltered_channels_are_tab0(oracle):
    img = np.full((4, 12), 200, np.uint8)
    sob, raw = SD.prefilter(img, 61)
    assert (sob[:, 0] == 61).all() and (sob[:, -1] == 61).all() and (raw[:, 0] == 61).all() and (raw[:, -1] == 61).all()
    assert (raw[:, 1:-1] == 200).all()
    # behavioural form on the C oracle: the last image-1 column's own grey value never reaches the cost of x = width1-1
    img1, img2 = _bt_images()
    op = oracle.SgbmParams(0, 16, 1, 8, 32, 1, -1, 60, -70, 16, 5)
    a = oracle.sgbm_compute(img1, img2, op, dump=True)[2]
    img1b = img1.copy(); img1b[:, 19] = 0          # Sobel at X=18 stays clipped at its value? 4*(0-110) = -440 -> 0, was 21
    b = oracle.sgbm_compute(img1b, img2, op, dump=True)[2]
    np.testing.assert_array_equal(b, SD.pixel_cost(img1b, img2, SD.derived(0, 16, 1, 8, 32, 1, -1, 60)))
    u_last = SD.prefilter(img1b, 61)
    assert u_last[0][0, 19] == 61 and u_last[1][0, 19] == 61 and a.shape == b.shape


def test_assumption_cost_buffer_carries_a_p2_bias_that_counts_toward_the_int16_range(oracle):
    right, left = synth.make_pair(64, 48, 16, frame_idx=1)
    R, L = _pad(right, left, 16)
    p = oracle.wass_params(16)
    _, st, Co, _, _ = oracle.sgbm_compute(R, L, p, dump=True)
    assert st.max_C == int(Co.max()) + p.P2


def test_assumption_horizontal_window_clamps_in_the_width1_domain(oracle):
    """The 3-wide window of output column 0 is pix(0) + pix(0) + pix(1): it replicates the first VALID column, it does not
    reach into the image columns left of minX1."""
    right, left = synth.make_pair(40, 1, 16, frame_idx=3)
    R, L = _pad(right, left, 16)
    pix = oracle.sgbm_compute(R, L, oracle.wass_params(16, win=1), dump=True)[2].astype(np.int64)
    C3 = oracle.sgbm_compute(R, L, oracle.wass_params(16, win=3), dump=True)[2].astype(np.int64)
    w1 = pix.shape[1]
    np.testing.assert_array_equal(C3[0, 0], 3 * (2 * pix[0, 0] + pix[0, 1]))           # one row: the 3 window rows coincide
    np.testing.assert_array_equal(C3[0, w1 - 1], 3 * (2 * pix[0, w1 - 1] + pix[0, w1 - 2]))
    np.testing.assert_array_equal(C3[0, 5], 3 * (pix[0, 4] + pix[0, 5] + pix[0, 6]))


def test_assumption_median_3x3_is_applied_unconditionally_after_aggregation(oracle):
    R, L = _textured(oracle)
    disp, _, _, _, raw = oracle.sgbm_compute(R, L, oracle.wass_params(16), dump=True)
    np.testing.assert_array_equal(disp, SD.median3(raw.astype(np.int64)).astype(np.int16))
    assert (disp != raw).any()


def test_assumption_3x3_solve_is_the_closed_form_cramer_rule(oracle):
    """cv::solve(DECOMP_LU) on a 3x3 system: the oracle's closed form must agree with an LU solve to rounding."""
    rng = np.random.default_rng(5)
    R = np.eye(3); T = np.array([1.0, 0.0, 0.0])
    for _ in range(20):
        P = rng.uniform([-2, -1, 5], [2, 1, 40])
        p = P[:2] / P[2]
        Q = R @ P + T
        q = Q[:2] / Q[2]
        np.testing.assert_allclose(oracle.triangulate_point(p, q, R, T), P, rtol=1e-9)
