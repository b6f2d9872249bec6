"""A net under the chain kernels that needs no GPU: the ISA hipcc emits for k_pair is EXECUTED by a small in-order interpreter
(scripts/gcn_interp.py: 64-lane waves, a workgroup's waves switched at barriers, the ~90 opcodes those kernels use) on random inputs, for two differently optimised builds of the
same source -- the shipped flags, and the same with the machine-sinking pass off.  Two correct builds compute the same S whatever
the inputs are; the round-3 miscompile of k_pair<2, 8, 1> (NOTES/traps.md: hand-over vectors prefetched into registers that the
loop body then uses as temporaries) shows up in this comparison on every chain of two or more full segments, while the build with
the sinking pass off agrees with the fenced build -- the second test rebuilds that source from the history and checks exactly that."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))


def _listing(src, out, extra=()):
    from wass_amd import build
    flags = [f for f in build.FLAGS if f not in ("-fPIC", "-Wall")]
    cmd = [build.HIPCC, *flags, *extra, "--cuda-device-only", "-S", src, "-o", out]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    return out


def _compare(a, b, prefix, cases):
    import gcn_interp as g
    out = []
    for (geom, chain, end) in cases:
        np_ = int(prefix.split("k_pairILi")[1][0])
        sa, _ = g.pair_case(a, prefix, *geom, chain, 4242 + chain, with_endstate=end, NP=np_)
        sb, _ = g.pair_case(b, prefix, *geom, chain, 4242 + chain, with_endstate=end, NP=np_)
        out.append(int((sa != sb).sum()))
    return out


# (width1, h, dx, dy), chain, split family: chains of four and of two full segments (+ tails)
CASES = [((40, 36, 1, 1), 2, False), ((40, 36, 1, 1), 23, True)]


@pytest.fixture(scope="module")
def listings(tmp_path_factory):
    """sgm_aggregate.hip compiled to ISA text twice: the shipped flags, and the same with the machine-sinking pass off"""
    if not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("no hipcc")
    d = tmp_path_factory.mktemp("isa")
    src = os.path.join(ROOT, "wass_amd", "csrc", "sgm_aggregate.hip")
    with ThreadPoolExecutor(2) as ex:
        fa = ex.submit(_listing, src, str(d / "shipped.s"))
        fb = ex.submit(_listing, src, str(d / "nosink.s"), ("-mllvm", "-disable-machine-sink"))
        return fa.result(), fb.result()


def test_two_builds_of_the_shipped_chain_kernels_compute_the_same(listings):
    a, b = listings
    for smode in (0, 1):
        diffs = _compare(a, b, "_ZN4wass6k_pairILi2ELi8ELi%dEEE" % smode, CASES)
        assert diffs == [0] * len(CASES), (smode, diffs)
    # the forward sweep of a family (checkpoints, minima, end states) and the unpaired paths of the 5-path mode
    import gcn_interp as g
    for fn, prefix, kw in ((g.ckpt_case, "_ZN4wass6k_ckptILi2ELi8EEE", {}), (g.ckpt_case, "_ZN4wass6k_ckptILi2ELi8EEE", {"with_endstate": True}),
                           (g.sweep_case, "_ZN4wass7k_sweepILi2ELi0ELi8EEE", {}), (g.sweep_case, "_ZN4wass7k_sweepILi2ELi1ELi8EEE", {})):
        ra, _ = fn(a, prefix, 40, 36, 1, 1, 2, 99, **kw)
        rb, _ = fn(b, prefix, 40, 36, 1, 1, 2, 99, **kw)
        assert ra.any() and int((ra != rb).sum()) == 0, (prefix, kw)
    # D = 512 (config E): four packed pairs per lane
    assert _compare(a, b, "_ZN4wass6k_pairILi4ELi8ELi1EEE", CASES[:1]) == [0]
    # paths 0 and 4 over one row (entry states per block of columns, minima); a width that leaves a partial block
    ra, _ = g.rowsweep_case(a, "_ZN4wass10k_rowsweepILi2EEE", 43, 8, 5, 3)
    rb, _ = g.rowsweep_case(b, "_ZN4wass10k_rowsweepILi2EEE", 43, 8, 5, 3)
    assert ra.any() and int((ra != rb).sum()) == 0
    # columns + rows in one kernel: a workgroup of ten waves with its barriers (a full block of the upper half, the partial block of the lower)
    base = None
    for block in (0, 5):
        ra, _ = g.pairx_case(a, "_ZN4wass7k_pairxILi2ELi8ELb1ELb0EEE", 23, 44, block, 5)
        rb, _ = g.pairx_case(b, "_ZN4wass7k_pairxILi2ELi8ELb1ELb0EEE", 23, 44, block, 5)
        assert int((ra != rb).sum()) == 0, block
        assert base is None or (ra != base).any()           # (the two workgroups wrote different pixels: the kernel did run)
        base = ra


def test_the_comparison_sees_the_round_3_miscompile(tmp_path):
    """The round-3 source (4985bc4) without the compiler fence in Rec::load against the same with the sinking pass off."""
    if not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("no hipcc")
    csrc = tmp_path / "a" / "b" / "csrc"
    csrc.mkdir(parents=True)
    (tmp_path / "a" / "include").mkdir()
    def show(path):
        r = subprocess.run(["git", "-C", ROOT, "show", "4985bc4:" + path], capture_output=True)
        if r.returncode != 0:
            pytest.skip("no history here")
        return r.stdout
    for f in ("sgm_aggregate.hip", "sgm_step.h", "common.h"):
        (csrc / f).write_bytes(show("wass_amd/csrc/" + f))
    (tmp_path / "a" / "include" / "wass_gpu.h").write_bytes(show("include/wass_gpu.h"))
    text = (csrc / "sgm_aggregate.hip").read_text()
    assert 'asm volatile("" ::: "memory");' in text
    (csrc / "nofence.hip").write_text(text.replace('asm volatile("" ::: "memory");', "/* no fence */", 1))
    with ThreadPoolExecutor(3) as ex:
        f0 = ex.submit(_listing, str(csrc / "sgm_aggregate.hip"), str(tmp_path / "fence.s"))
        f1 = ex.submit(_listing, str(csrc / "nofence.hip"), str(tmp_path / "nofence.s"))
        f2 = ex.submit(_listing, str(csrc / "nofence.hip"), str(tmp_path / "nofence_nosink.s"), ("-mllvm", "-disable-machine-sink"))
        fence, nofence, nosink = f0.result(), f1.result(), f2.result()
    P = "_ZN4wass6k_pairILi2ELi8ELi1EEE"
    cases = [((40, 36, 1, 1), 20, False), ((40, 36, 1, 1), 26, False)]      # two full segments; one full segment
    assert _compare(fence, nosink, P, cases) == [0, 0]
    bad = _compare(fence, nofence, P, cases)
    assert bad[0] > 0 and bad[1] == 0, bad


def _sgm_family_model(C, dx, dy, P1, P2, both=True):
    """L_r(p, d) = C(p, d) + min(L_r(p - r, d), L_r(p - r, d +- 1) + P1, min_k L_r(p - r, k) + P2) - min_k L_r(p - r, k), zero state outside the
    image, for r = (dx, dy) and its opposite; returns their sum"""
    h, w, D = C.shape
    S = np.zeros(C.shape, np.int64)
    big = 1 << 20
    for sx, sy in (((dx, dy), (-dx, -dy)) if both else ((dx, dy),)):
        L = np.zeros(C.shape, np.int64)
        for y in (range(h) if sy > 0 else range(h - 1, -1, -1)):
            for x in (range(w) if sx >= 0 else range(w - 1, -1, -1)):
                px, py = x - sx, y - sy
                prev = L[py, px] if 0 <= px < w and 0 <= py < h else np.zeros(D, np.int64)
                m = prev.min()
                lo = np.concatenate(([big], prev[:-1])) + P1
                hi = np.concatenate((prev[1:], [big])) + P1
                L[y, x] = C[y, x] + np.minimum(np.minimum(prev, lo), np.minimum(hi, m + P2)) - m
        S += L
    return S


def test_the_shipped_isa_computes_a_diagonal_family_like_the_recurrence(listings):
    """Not a comparison of two builds but of the ISA with the algorithm: k_ckpt<2, 8> and k_pair<2, 8, 0> as hipcc emits them, run by the
    interpreter over every chain of a small image, against the path recurrence written out in numpy -- every cell of S."""
    import gcn_interp as g
    a = listings[0]
    for (w1, h, dx, dy) in ((26, 30, 1, 1), (21, 12, -1, 1)):           # chains of up to three full segments + tail; the anti-diagonals
        C, S = g.family_case(a, w1, h, dx, dy, 11)
        M = _sgm_family_model(C.astype(np.int64), dx, dy, 7, 150)
        assert M.max() < 32767
        assert np.array_equal(S.astype(np.int64), M), (w1, h, dx, dy, int((S.astype(np.int64) != M).sum()))
    # the accumulating form, and the LAST family's: S is read, finished with one saturation at 0x7FFF and (debug fetch) stored again --
    # the selection code runs behind it
    for smode, smax in ((1, 3000), (2, 31000)):
        C, S, S0 = g.family_case(a, 21, 27, -1, 1, 5, smode=smode, smax=smax)
        M = S0.astype(np.int64) + _sgm_family_model(C.astype(np.int64), -1, 1, 7, 150)
        if smode == 2:
            assert (M > 0x7FFF).any()
            M = np.minimum(M, 0x7FFF)
        assert np.array_equal(S.astype(np.int64), M), (smode, int((S.astype(np.int64) != M).sum()))


def test_the_shipped_isa_computes_columns_and_rows_like_the_recurrence(listings):
    """The fused half of the 8-path schedule as the host launches it -- k_rowsweep<2> (entry states, minima), k_ckpt<2, 8> over the split
    column family, k_pairx<2, 8, true, false> in workgroups of ten waves with their barriers -- on a small image with a partial block
    of columns and tail segments in both halves: S grows by exactly L_0 + L_4 + L_2 + L_6 of the recurrence, every cell."""
    import gcn_interp as g
    C, S0, S1 = g.columns_rows_case(listings[0], 23, 44, 3)
    Ci = C.astype(np.int64)
    M = S0.astype(np.int64) + _sgm_family_model(Ci, 0, 1, 7, 150) + _sgm_family_model(Ci, 1, 0, 7, 150)
    assert M.max() < 32767
    assert np.array_equal(S1.astype(np.int64), M), int((S1.astype(np.int64) != M).sum())


def test_the_shipped_isa_computes_path_2_and_the_rows_of_the_5_path_mode_like_the_recurrence(listings):
    """Round 6, 5-path mode (MODE_SGBM, what the reference runs): the same walk with k_pairx<2, 8, true, ONE = true> -- of the column family only
    the downward path counts (forward recomputation in the upper half of the image, backward path in the lower half, which starts from the
    upper half's end state); S grows by exactly L_2 + L_0 + L_4."""
    import gcn_interp as g
    C, S0, S1 = g.columns_rows_case(listings[0], 23, 44, 4, one=True)
    Ci = C.astype(np.int64)
    M = S0.astype(np.int64) + _sgm_family_model(Ci, 0, 1, 7, 150, both=False) + _sgm_family_model(Ci, 1, 0, 7, 150)
    assert M.max() < 32767
    assert np.array_equal(S1.astype(np.int64), M), int((S1.astype(np.int64) != M).sum())


def test_the_shipped_isa_computes_an_unpaired_path_like_the_recurrence(listings):
    """5-path mode (what the reference runs): paths 1 and 3 have no partner and go through k_sweep -- the accumulating form on a diagonal, the
    last form (saturation, stored for the debug fetch, selection behind it) on the other."""
    import gcn_interp as g
    for dx, dy, smode, smax in ((1, 1, 1, 3000), (-1, 1, 2, 31500)):
        C, S0, S1 = g.single_path_case(listings[0], 22, 19, dx, dy, 9, smode=smode, smax=smax)
        M = S0.astype(np.int64) + _sgm_family_model(C.astype(np.int64), dx, dy, 7, 150, both=False)
        if smode == 2:
            assert (M > 0x7FFF).any()
            M = np.minimum(M, 0x7FFF)
        assert np.array_equal(S1.astype(np.int64), M), (dx, dy, smode, int((S1.astype(np.int64) != M).sum()))
