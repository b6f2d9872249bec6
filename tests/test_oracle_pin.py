"""The pin for rows a2-a6: oracle/sgbm_oracle.c against the REAL cv::StereoSGBM.

tests/golden/sgbm_opencv.npz is written by scripts/pin_with_opencv.py on a machine that has OpenCV (the build image has
none: no cv2, no library, and the reference cannot be compiled here -- SURVEY.md 8c).  While the file is absent this test
is skipped and parity of the SGBM stage stays "unpinned"; with the file present every stored map must be reproduced bit
for bit, MODE_SGBM and MODE_HH, and the test that checks the cases the script itself runs keeps script and test in step.
"""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PIN = os.path.join(ROOT, "tests", "golden", "sgbm_opencv.npz")


def _cases():
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import pin_with_opencv
    return pin_with_opencv.cases()


def test_pin_script_cases_run_through_the_oracle(oracle):
    """Every case the pin script would hand to OpenCV goes through the oracle (both modes) without an error, and only the
    overflow probe leaves the int16 range -- so a maintainer's pin file will contain comparable maps."""
    names = set()
    for c in _cases():
        assert c["name"] not in names
        names.add(c["name"])
        for mode in (5, 8):
            p = oracle.wass_params(c["D"], mode, min_disp=c["mind"], win=c["win"], p1_mult=c["p1"], p2_mult=c["p2"])
            d, st = oracle.dense_disparity16(c["right"], c["left"], p, disparity_offset=c["off"])
            assert d.shape == c["right"].shape
            assert bool(st.overflow) == (c["name"] == "overflow_probe"), c["name"]


@pytest.mark.skipif(not os.path.exists(PIN), reason="tests/golden/sgbm_opencv.npz not generated (scripts/pin_with_opencv.py needs cv2)")
def test_oracle_reproduces_opencv(oracle):
    z = np.load(PIN)
    bad = []
    for name in [str(n) for n in z["names"]]:
        D, win, mind, p1, p2, off = (int(v) for v in z[f"{name}__cfg"])
        for key, mode in (("sgbm", 5), ("hh", 8)):
            p = oracle.wass_params(D, mode, min_disp=mind, win=win, p1_mult=p1, p2_mult=p2)
            d, st = oracle.dense_disparity16(z[f"{name}__right"], z[f"{name}__left"], p, disparity_offset=off)
            if st.overflow:
                continue            # A.7: OpenCV itself is build-dependent there (scalar wraps, SIMD saturates); recorded, not compared
            if not np.array_equal(d, z[f"{name}__{key}"]):
                bad.append((name, key, int((d != z[f"{name}__{key}"]).sum())))
    assert not bad, f"oracle differs from OpenCV {z['opencv_version']}: {bad}"
