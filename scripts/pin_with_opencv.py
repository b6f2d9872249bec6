#!/usr/bin/env python3
"""Pin the SGBM oracle to the real cv::StereoSGBM -- for a machine that HAS OpenCV (the build image does not).

    python scripts/pin_with_opencv.py            # writes tests/golden/sgbm_opencv.npz
    python -m pytest tests/test_oracle_pin.py    # compares oracle/sgbm_oracle.c with it (skipped while the file is absent)

The reference computes its disparity with cv::StereoSGBM (wass_stereo/wass_stereo.cpp:775-782, compute() at :837; OpenCV
4.5.5 per meta.yaml:12-13).  OpenCV is neither vendored in the reference tree nor installed in the image this repository
was built in, so oracle/sgbm_oracle.c restates the published algorithm and every "bit-exact" claim
for rows a2-a6 is exact against that restatement.  This script closes the loop: it runs the real library with the
reference's parameters and padding on the inputs the tests use and stores inputs + outputs; with the file present,
tests/test_oracle_pin.py requires the oracle to reproduce every map, mode by mode.

What is run, per case:  StereoSGBM_create(minDisparity, numDisparities, blockSize = WINSIZE, P1, P2) followed by the
setters of wass_stereo.cpp:778-782 (uniquenessRatio 1, disp12MaxDiff -1, preFilterCap 60, speckleRange 16,
speckleWindowSize -70) in MODE_SGBM (what the reference runs) and MODE_HH (the commented-out fullDP switch, :777; the
8-path mode of the roofline target); the images are padded exactly as :820-839 do and the result is cropped the same way.
Also recorded: one input that drives block cost + P2 beyond int16 -- there OpenCV's scalar build wraps and
its SIMD build saturates, so the stored map tells which of the two the maintainer's build is (the tests only compare it
when the oracle flags no overflow).
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def pad(right, left, D, off=0):                     # wass_stereo.cpp:801-831
    h, w = right.shape
    offp, comp = max(off, 0), max(-off, 0)
    Wp = w + D + offp
    R = np.zeros((h, Wp), np.uint8)
    L = np.zeros((h, Wp), np.uint8)
    R[:, D:D + w] = right
    L[:, D + offp - comp:D + offp - comp + w] = left
    return R, L


def cases():
    from wass_amd import synth
    out = []
    z = np.load(os.path.join(ROOT, "tests", "golden", "sgbm_regress.npz"))
    for name in "abc":                               # the regression pairs the oracle tests already hold
        w, h, D, mode = (int(v) for v in z[f"{name}_cfg"])
        out.append(dict(name=f"regress_{name}", right=z[f"{name}_right"], left=z[f"{name}_left"], D=D, win=13, mind=1, p1=2, p2=64, off=0))
    for (w, h, D, win, mind, off) in ((160, 120, 32, 13, 1, 0), (200, 90, 64, 9, 1, 0), (131, 77, 80, 7, 1, 3), (131, 77, 80, 7, 1, -4),
                                      (320, 64, 256, 13, 1, 0), (150, 70, 128, 5, 0, 0), (33, 29, 16, 3, 1, 0)):
        r, l = synth.make_pair(w, h, D, frame_idx=w + h + D)
        out.append(dict(name=f"synth_{w}x{h}_D{D}_w{win}_m{mind}_o{off}", right=r, left=l, D=D, win=win, mind=mind, p1=2, p2=64, off=off))
    rng = np.random.default_rng(20260929)            # Appendix F probes: ties, rejections, borders
    noise = rng.integers(0, 256, (40, 120), dtype=np.uint8)
    out.append(dict(name="probe_noise", right=noise, left=rng.integers(0, 256, (40, 120), dtype=np.uint8), D=32, win=13, mind=1, p1=2, p2=64, off=0))
    out.append(dict(name="probe_shift5", right=noise, left=np.roll(noise, 5, axis=1), D=32, win=5, mind=1, p1=2, p2=64, off=0))
    out.append(dict(name="probe_constant", right=np.full((30, 90), 128, np.uint8), left=np.full((30, 90), 128, np.uint8), D=16, win=13, mind=1, p1=2, p2=64, off=0))
    binary = (rng.integers(0, 2, (40, 100)) * 255).astype(np.uint8)   # A.7: block cost + P2 > 32767
    out.append(dict(name="overflow_probe", right=binary, left=(rng.integers(0, 2, (40, 100)) * 255).astype(np.uint8), D=16, win=17, mind=1, p1=2, p2=100, off=0))
    return out


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden", "sgbm_opencv.npz"))
    args = ap.parse_args()
    try:
        import cv2
    except ImportError:
        sys.exit("pin_with_opencv.py needs OpenCV's Python module (cv2, ideally 4.5.x as the reference pins it); "
                 "it is not installed here.  Nothing written.")
    ver = cv2.__version__
    if not ver.startswith("4.5"):
        print(f"warning: cv2 {ver}; the reference pins OpenCV 4.5.5 (meta.yaml:12-13)", file=sys.stderr)
    store = {"opencv_version": np.array(ver), "build_info_simd": np.array(cv2.getBuildInformation().split("CPU/HW features")[-1][:400])}
    names = []
    for c in cases():
        win, D = c["win"], c["D"]
        R, L = pad(c["right"], c["left"], D, c["off"])
        w = c["right"].shape[1]
        for mode_name, mode, nd in (("sgbm", cv2.STEREO_SGBM_MODE_SGBM, 5), ("hh", cv2.STEREO_SGBM_MODE_HH, 8)):
            s = cv2.StereoSGBM_create(c["mind"], D, win, c["p1"] * win * win, c["p2"] * win * win)      # wass_stereo.cpp:775
            s.setMode(mode)
            s.setUniquenessRatio(1); s.setDisp12MaxDiff(-1); s.setPreFilterCap(60)                      # :778-782
            s.setSpeckleRange(16); s.setSpeckleWindowSize(-70)
            disp = s.compute(R, L)[:, D:D + w]                                                           # :837-839
            store[f"{c['name']}__{mode_name}"] = disp.astype(np.int16)
        store[f"{c['name']}__right"] = c["right"]
        store[f"{c['name']}__left"] = c["left"]
        store[f"{c['name']}__cfg"] = np.array([D, win, c["mind"], c["p1"], c["p2"], c["off"]], np.int64)
        names.append(c["name"])
    store["names"] = np.array(names)
    np.savez_compressed(args.out, **store)
    print(f"wrote {args.out}: {len(names)} cases x 2 modes, OpenCV {ver}")


if __name__ == "__main__":
    main()
