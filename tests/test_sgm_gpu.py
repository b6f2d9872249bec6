"""GPU parity of the SGBM stages against the CPU oracle.

Bit-exact: every stage is integer arithmetic.  All calls go through the C ABI.
"""
import os
import numpy as np
import pytest

import wass_amd
from wass_amd import default_sgm_params, synth

pytestmark = pytest.mark.gpu


def _oracle_params(O, p):
    return O.SgbmParams(p.min_disp, p.num_disp, p.win, p.P1, p.P2, p.uniq_ratio, p.disp12_max_diff,
                        p.prefilter_cap, p.speckle_win, p.speckle_range, p.ndirs)


def _pad(right, left, D, off=0):
    h, w = right.shape
    offp, comp = max(off, 0), max(-off, 0)
    Wp = w + D + offp
    R = np.zeros((h, Wp), np.uint8); L = np.zeros((h, Wp), np.uint8)
    R[:, D:D + w] = right
    L[:, D + offp - comp:D + offp - comp + w] = left
    return R, L


CASES = [
    # w, h, D, ndirs, win, min_disp, off
    (64, 48, 16, 5, 13, 1, 0),
    (64, 48, 16, 8, 13, 1, 0),
    (160, 120, 32, 5, 13, 1, 0),
    (160, 120, 32, 8, 13, 1, 0),
    (200, 90, 64, 5, 13, 1, 0),
    (200, 90, 64, 8, 9, 1, 0),
    (150, 70, 128, 8, 13, 1, 0),
    (150, 70, 128, 5, 5, 0, 0),
    (131, 77, 48, 8, 13, 2, 0),       # ragged sizes, D not a multiple of 64, minD = 2
    (131, 77, 80, 5, 7, 1, 3),        # positive DISPARITY_OFFSET
    (131, 77, 80, 8, 7, 1, -4),       # negative DISPARITY_OFFSET
    (320, 64, 256, 8, 13, 1, 0),
    (300, 40, 160, 5, 13, 1, 0),      # NP = 2 with padded slots
    (340, 32, 272, 8, 13, 1, 0),      # NP = 3
    (330, 40, 272, 5, 13, 1, 0),
    (560, 24, 512, 8, 13, 1, 0),      # NP = 4
    (700, 20, 640, 5, 13, 1, 0),      # NP = 5: the WASS default MAX_DISPARITY
    (40, 300, 16, 8, 13, 1, 0),       # tall and narrow: long columns, short rows
    (33, 29, 16, 8, 3, 1, 0),         # tiny
    (900, 12, 768, 8, 13, 1, 0),      # NP = 6
    (1100, 10, 1024, 5, 13, 1, 0),    # NP = 8: the largest supported MAX_DISPARITY
    (257, 33, 96, 8, 11, 1, 0),       # (wider windows leave the int16 range of A.7 on textured input)
    (75, 41, 32, 5, 11, 3, 0),        # odd width, minD = 3, window wider than a checkpoint segment is tall
    # every checkpoint regime with half chains of at least two full segments (F >= 2: the steady-state loop of k_pair runs for
    # the column and diagonal families too), every NP from 4 up in 8-path mode
    (600, 64, 512, 8, 13, 1, 0),      # NP = 4, K = 8
    (720, 40, 640, 8, 13, 1, 0),      # NP = 5, K = 4
    (800, 40, 768, 8, 13, 1, 0),      # NP = 6
    (950, 40, 896, 8, 13, 1, 0),      # NP = 7 (no other case has it)
    (1100, 40, 1024, 8, 13, 1, 0),    # NP = 8, 8-path
    (460, 48, 384, 8, 13, 1, 0),      # NP = 3, K = 8, three segments per half chain
    (460, 48, 384, 5, 13, 1, 0),
    (950, 40, 896, 5, 13, 1, 0),
]


@pytest.mark.parametrize("w,h,D,ndirs,win,mind,off", CASES)
def test_stage_parity(gpu_ctx, oracle, w, h, D, ndirs, win, mind, off):
    right, left = synth.make_pair(w, h, D, frame_idx=w + h + D)
    p = default_sgm_params(D, ndirs=ndirs, win=win, min_disp=mind, disp_offset=off)
    gpu_ctx.set_debug(True)
    try:
        got = gpu_ctx.sgm_disparity(right, left, p)
        Cg, Sg, rawg = gpu_ctx.sgm_debug_fetch(w, h, p)
    finally:
        gpu_ctx.set_debug(False)
    # production mode (S never written) must give the same map
    np.testing.assert_array_equal(gpu_ctx.sgm_disparity(right, left, p), got)

    R, L = _pad(right, left, D, off)
    disp, st, Co, So, rawo = oracle.sgbm_compute(R, L, _oracle_params(oracle, p), dump=True)
    assert not st.overflow
    assert Cg.shape == Co.shape
    np.testing.assert_array_equal(Cg, Co, err_msg="cost volume C")
    np.testing.assert_array_equal(Sg, So, err_msg="aggregated volume S")
    np.testing.assert_array_equal(rawg, rawo, err_msg="raw disparity (WTA/uniqueness/subpixel/LR)")
    np.testing.assert_array_equal(got, disp[:, D:D + w], err_msg="median + crop")
    # and the wass-level wrapper of the oracle agrees with itself
    d2, _ = oracle.dense_disparity16(right, left, _oracle_params(oracle, p), off)
    np.testing.assert_array_equal(got, d2)


# Chain lengths around the checkpoint distance: an image of height h has column half-chains of h // 2 and h - h // 2 rows and
# diagonal chains of every length 1 .. min(w, h), so heights 1 .. 2K + 1 (and a few beyond) put every combination of
# "number of full segments F" and "tail length r" of k_ckpt / k_pair / k_pairx / k_sweep through the guarded prologues and
# epilogues, for every NP (= every template instance that is launched) in both modes.  One known instance-specific
# miscompile (k_pair<2, 8, 1>) showed only in segment 0 of short diagonal chains.
_SHORT_H = [1, 2, 3, 4, 5, 7, 8, 9, 11, 15, 16, 17, 18, 23, 25, 33, 34]


@pytest.mark.parametrize("ndirs", [5, 8])
@pytest.mark.parametrize("D", [64, 128, 256, 384, 512, 640, 768, 896, 1024])
def test_short_chains_every_instance(gpu_ctx, oracle, D, ndirs):
    w = 48 + (D % 7)                                       # width1 = w + D - 1 ... a few blocks of ten columns plus a ragged one
    gpu_ctx.set_debug(True)
    try:
        for h in _SHORT_H:
            right, left = synth.make_pair(w, h, D, frame_idx=h * 31 + D)
            p = default_sgm_params(D, ndirs=ndirs, win=5)
            got = gpu_ctx.sgm_disparity(right, left, p, allow_overflow=True)
            Cg, Sg, rawg = gpu_ctx.sgm_debug_fetch(w, h, p)
            R, L = _pad(right, left, D)
            disp, st, Co, So, rawo = oracle.sgbm_compute(R, L, _oracle_params(oracle, p), dump=True)
            assert not st.overflow
            np.testing.assert_array_equal(Cg, Co, err_msg=f"C, h={h}")
            np.testing.assert_array_equal(Sg, So, err_msg=f"S, h={h}")
            np.testing.assert_array_equal(rawg, rawo, err_msg=f"raw, h={h}")
            np.testing.assert_array_equal(got, disp[:, D:D + w], err_msg=f"final, h={h}")
    finally:
        gpu_ctx.set_debug(False)


@pytest.mark.parametrize("ndirs", [5, 8])
@pytest.mark.parametrize("D", [64, 256, 384, 512, 640, 768, 896, 1024])
def test_device_selftest_passes(gpu_ctx, D, ndirs):
    """wass_sgm_selftest: the production schedule against one plain sweep per path, compared on the device (no oracle).  It
    must pass on a healthy build for every NP; scripts/selftest.py shows it failing on the build without the compiler fence
    of Rec::load (profiles/r04_selftest_nofence.txt)."""
    for (w, h) in ((D + 56, 40), (D + 40, 17)):
        assert gpu_ctx.sgm_selftest(w, h, D, ndirs) == 0


def test_random_noise_images(gpu_ctx, oracle):
    """Untextured/random inputs: many ties, rejected pixels and saturated S."""
    rng = np.random.default_rng(7)
    w, h, D = 140, 60, 32
    right = rng.integers(0, 256, (h, w), dtype=np.uint8)
    left = rng.integers(0, 256, (h, w), dtype=np.uint8)
    for ndirs in (5, 8):
        p = default_sgm_params(D, ndirs=ndirs, p2_mult=16)
        got = gpu_ctx.sgm_disparity(right, left, p, allow_overflow=True)
        d2, st = oracle.dense_disparity16(right, left, _oracle_params(oracle, p))
        # inside the int16 range of A.7 the maps are equal; outside it the reference itself depends on its build (OpenCV's
        # scalar code wraps, its SIMD code saturates) and what is required is that the GPU SAYS so, exactly when the oracle does
        assert gpu_ctx.sgm_timings().cost_overflow == int(bool(st.overflow))
        if not st.overflow:
            np.testing.assert_array_equal(got, d2)


def test_constant_images(gpu_ctx, oracle):
    """KAT (i) of SURVEY.md 8c: all costs zero -> d = 0 everywhere valid."""
    c = np.full((40, 100), 77, np.uint8)
    for ndirs in (5, 8):
        p = default_sgm_params(16, ndirs=ndirs)
        got = gpu_ctx.sgm_disparity(c, c, p)
        d2, _ = oracle.dense_disparity16(c, c, _oracle_params(oracle, p))
        np.testing.assert_array_equal(got, d2)
        assert set(np.unique(got)) <= {0, 16}


def test_integer_shift(gpu_ctx):
    """KAT (ii): right(x) = left(x - k) -> raw disparity 16*k in the interior."""
    rng = np.random.default_rng(1)
    w, h, D, k = 160, 60, 32, 7
    left = rng.integers(1, 255, (h, w), dtype=np.uint8)
    right = np.zeros_like(left); right[:, k:] = left[:, :w - k]
    for ndirs in (5, 8):
        got = gpu_ctx.sgm_disparity(right, left, default_sgm_params(D, ndirs=ndirs))
        assert (got[10:-10, 40:-10] == 16 * k).all()


def test_device_pointer_entry(gpu_ctx, oracle):
    import torch
    w, h, D = 160, 120, 32
    right, left = synth.make_pair(w, h, D, frame_idx=3)
    p = default_sgm_params(D, ndirs=8)
    dr = torch.from_numpy(right).cuda(); dl = torch.from_numpy(left).cuda()
    out = gpu_ctx.sgm_disparity_dev(dr, dl, p)
    gpu_ctx.synchronize()
    d2, _ = oracle.dense_disparity16(right, left, _oracle_params(oracle, p))
    np.testing.assert_array_equal(out.cpu().numpy(), d2)
    t = gpu_ctx.sgm_timings()
    assert t.total_ms > 0 and t.aggregate_launches >= 1 and t.cost_overflow == 0


@pytest.mark.parametrize("seed", range(6))
def test_random_small_shapes_match_the_oracle(gpu_ctx, oracle, seed):
    """Heights and widths down to a few pixels: chains of length 1-3, families split in the middle with an empty half, windows
    wider than the image, rows fewer than a checkpoint segment -- final map and S volume against the oracle, both path modes."""
    rng = np.random.default_rng(1000 + seed)
    noverflow = 0
    for _ in range(6):
        h = int(rng.integers(1, 24)); w = int(rng.integers(3, 60)); D = int(rng.choice([16, 32, 48]))
        win = int(rng.choice([1, 3, 5, 9, 13])); ndirs = int(rng.choice([5, 8])); mind = int(rng.integers(0, 3))
        right = rng.integers(0, 256, (h, w), dtype=np.uint8)
        left = np.roll(right, int(rng.integers(1, 6)), axis=1) if rng.random() < 0.7 else rng.integers(0, 256, (h, w), dtype=np.uint8)
        p = default_sgm_params(D, ndirs=ndirs, win=win, min_disp=mind, p2_mult=int(rng.choice([32, 8])))
        p.uniq_ratio = int(rng.choice([0, 1, 10]))
        try:
            ref, st = oracle.dense_disparity16(right, left, _oracle_params(oracle, p))
        except RuntimeError:                                      # image narrower than half the window: OpenCV reads past the row there
            with pytest.raises(Exception):
                gpu_ctx.sgm_disparity(right, left, p)
            continue
        if st.overflow:                                           # outside the int16 precondition of A.7: the call must say so
            with pytest.raises(wass_amd.WassError) as e:
                gpu_ctx.sgm_disparity(right, left, p)
            assert e.value.code == -5
            noverflow += 1
            continue
        got = gpu_ctx.sgm_disparity(right, left, p)
        np.testing.assert_array_equal(got, ref, err_msg=f"w={w} h={h} D={D} win={win} ndirs={ndirs} minD={mind}")


def test_uploads_are_ordered_before_the_calls_that_read_them(gpu_ctx):
    """wass_upload_async copies on the context's copy stream; wass_burned_area_mask_dev and wass_sgm_disparity_dev wait for
    the uploads that cover their inputs.  Several frames are uploaded back to back into a ring of buffers (as bench.py does,
    one frame ahead) and every result must equal the one computed from inputs that were resident all along."""
    import torch
    w, h, D = 320, 200, 32
    p = default_sgm_params(D, ndirs=8)
    frames = [synth.make_pair(w, h, D, frame_idx=40 + k) for k in range(5)]
    for r, _ in frames:
        r[5:9, 7:30] = 255                                        # burned pixels for the mask
    want = [gpu_ctx.sgm_disparity(r, l, p) for r, l in frames]
    pinned = [(torch.from_numpy(r).pin_memory(), torch.from_numpy(l).pin_memory()) for r, l in frames]
    ring = [tuple(torch.zeros((h, w), dtype=torch.uint8, device="cuda") for _ in range(3)) for _ in range(3)]
    outs = [torch.empty((h, w), dtype=torch.int16, device="cuda") for _ in frames]
    torch.cuda.synchronize()

    def upload(k):
        gpu_ctx.upload_async(ring[k % 3][0], pinned[k][0])
        gpu_ctx.upload_async(ring[k % 3][1], pinned[k][1])
    upload(0)
    for k in range(len(frames)):
        dr, dl, dm = ring[k % 3]
        gpu_ctx.burned_area_mask_dev(dr, dm)
        if k + 1 < len(frames):
            upload(k + 1)
        gpu_ctx.sgm_disparity_dev(dr, dl, p, outs[k])
        if k == 2:
            gpu_ctx.synchronize()
            np.testing.assert_array_equal(dm.cpu().numpy(), (frames[k][0] <= 254).astype(np.uint8))
    gpu_ctx.synchronize()
    for k in range(len(frames)):
        np.testing.assert_array_equal(outs[k].cpu().numpy(), want[k])


def test_overflow_is_reported(gpu_ctx):
    """Costs beyond the int16 precondition (A.7) are flagged, not silently wrong."""
    import wass_amd
    rng = np.random.default_rng(3)
    w, h, D = 120, 50, 16
    right = (rng.integers(0, 2, (h, w)) * 255).astype(np.uint8)
    left = (rng.integers(0, 2, (h, w)) * 255).astype(np.uint8)
    p = default_sgm_params(D, ndirs=5, win=17, p2_mult=100)
    with pytest.raises(wass_amd.WassError) as e:
        gpu_ctx.sgm_disparity(right, left, p)
    assert e.value.code == -5


def test_random_parameter_sets_match_the_oracle(gpu_ctx, oracle):
    """SGBM parameters away from the WASS defaults (uniqueness ratio, disp12MaxDiff, pre-filter cap, P1/P2, window,
    minimum disparity, path count), drawn at random: final fixed-point disparity bit-exact against the oracle."""
    rng = np.random.default_rng(20260928)
    done = 0
    for trial in range(40):
        w, h = int(rng.integers(40, 180)), int(rng.integers(12, 70))
        D = int(rng.choice([16, 32, 48, 64, 96, 144]))
        win = int(rng.choice([3, 5, 7, 9, 11, 13]))
        mind = int(rng.integers(0, 4))
        p = default_sgm_params(D, ndirs=int(rng.choice([5, 8])), win=win, min_disp=mind)
        p.P1 = int(rng.integers(1, 40)) * win
        p.P2 = p.P1 + int(rng.integers(1, 400)) * win
        p.uniq_ratio = int(rng.choice([0, 1, 5, 10, 15, 40]))
        p.disp12_max_diff = int(rng.choice([-1, 0, 1, 2, 5]))
        p.prefilter_cap = int(rng.choice([5, 15, 31, 60, 63]))
        right, left = synth.make_pair(w, h, D, frame_idx=1000 + trial)
        ref, st = oracle.dense_disparity16(right, left, _oracle_params(oracle, p))
        if st.overflow:                                           # outside the int16 range where the reference is defined:
            with pytest.raises(wass_amd.WassError) as e:          # the GPU reports WASS_ERR_COST_OVERFLOW for exactly these inputs
                gpu_ctx.sgm_disparity(right, left, p)
            assert e.value.code == -5
            continue
        got = gpu_ctx.sgm_disparity(right, left, p)
        np.testing.assert_array_equal(got, ref, err_msg=f"trial {trial}: {w}x{h} D={D} win={win} minD={mind} P1={p.P1} P2={p.P2} "
                                                         f"uniq={p.uniq_ratio} d12={p.disp12_max_diff} cap={p.prefilter_cap} ndirs={p.ndirs}")
        done += 1
    assert done >= 25


# ---- round 6: the 5-path mode (MODE_SGBM, what the reference runs) on the fused kernel of the 8-path schedule: path 1's sweep writes S,
# k_pairx<.., ONE> adds path 2 and both row paths in one pass, path 3's sweep reads S and selects (S written twice instead of three
# times).  S bit-exact against the oracle for every NP the fused kernel is built for, for heights that put every combination of full
# segments / tail segments of the split column family through it, and widths with a partial block of columns.
_FUSED5_CASES = [(64, 48, 16), (160, 120, 32), (200, 90, 64), (150, 70, 128), (131, 77, 48), (320, 64, 256), (340, 32, 272), (560, 24, 512),
                 (600, 64, 512), (460, 48, 384), (40, 300, 16), (33, 29, 16), (257, 33, 96), (300, 37, 256), (290, 131, 192)]


@pytest.mark.parametrize("w,h,D", _FUSED5_CASES)
def test_fused_5path_schedule_is_bit_exact(gpu_ctx, oracle, w, h, D):
    right, left = synth.make_pair(w, h, D, frame_idx=w + h + D)
    p = default_sgm_params(D, ndirs=5)
    gpu_ctx.set_debug(True)
    try:
        got = gpu_ctx.sgm_disparity(right, left, p)
        Cg, Sg, rawg = gpu_ctx.sgm_debug_fetch(w, h, p)
    finally:
        gpu_ctx.set_debug(False)
    np.testing.assert_array_equal(gpu_ctx.sgm_disparity(right, left, p), got)          # production mode (S not kept)
    R, L = _pad(right, left, D, 0)
    disp, st, Co, So, rawo = oracle.sgbm_compute(R, L, _oracle_params(oracle, p), dump=True)
    assert not st.overflow
    np.testing.assert_array_equal(Sg, So, err_msg="aggregated volume S (path 2 + rows in k_pairx<.., ONE>)")
    np.testing.assert_array_equal(got, disp[:, D:D + w])


def test_kernel_events_report_the_schedule(gpu_ctx):
    """wass_ctx_set_kernel_events / wass_sgm_kernel_times (bench.py's kernel_ms): the launches of one SGM call by name, in launch order."""
    w, h, D = 320, 64, 256
    right, left = synth.make_pair(w, h, D, frame_idx=5)
    gpu_ctx.set_kernel_events(True)
    try:
        for ndirs, want in ((8, ["k_prefilter", "k_hsum_q", "k_vsum_col", "k_rowsweep", "k_ckpt(family 1)", "k_ckpt(family 2)", "k_pair(family 1)",
                                 "k_pairx", "k_pair(family 2)"]),
                            (5, ["k_prefilter", "k_hsum_q", "k_vsum_col", "k_rowsweep", "k_sweep(path 1)", "k_pairx", "k_sweep(path 3 + selection)"])):
            a = gpu_ctx.sgm_disparity(right, left, default_sgm_params(D, ndirs=ndirs))
            times = gpu_ctx.sgm_kernel_times()
            assert [n for n, _ in times] == want
            assert all(ms > 0 for _, ms in times)
    finally:
        gpu_ctx.set_kernel_events(False)
    np.testing.assert_array_equal(gpu_ctx.sgm_disparity(right, left, default_sgm_params(D, ndirs=5)), a)
