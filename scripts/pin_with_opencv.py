#!/usr/bin/env python3
"""Pin the oracle to the real OpenCV -- for a machine that HAS OpenCV (the build image does not).

    python scripts/pin_with_opencv.py            # writes tests/golden/sgbm_opencv.npz and tests/golden/opencv_other.npz
    python -m pytest tests/test_oracle_pin.py    # compares oracle/*.c with them (skipped while the files are absent)

One run pins every restatement of an OpenCV routine the hot path and its neighbours rest on: cv::StereoSGBM (rows a2-a6,
below), and -- other_cases() / run_other() -- stereoRectify + initUndistortRectifyMap + remap, warpPerspective (f1),
undistort and CLAHE (f2), resize, filterSpeckles and the Sobel / connected-component extraction (a9).

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


# ---------------------------------------------------------------------------------------------- the other OpenCV restatements
# Rows f1 (cv::stereoRectify + initUndistortRectifyMap + remap INTER_CUBIC, cv::warpPerspective: wass_stereo.cpp:515-516,541,
# 600-604), f2 (cv::undistort, cv::CLAHE: wass_prepare.cpp:36-39,257-275), a9 (cv::resize, cv::filterSpeckles, the Sobel /
# connectedComponents extraction: wass_stereo.cpp:788-796,903-904,947-986) and the previews of load_data (:413,416) are
# restated in oracle/rectify_oracle.c, clahe_oracle.c and a9_oracle.c.  other_cases() holds their inputs; run_other() is what
# the real library is asked for each of them.  Everything lands in tests/golden/opencv_other.npz.
def _rot(rx, ry, rz):
    cx, sx, cy, sy, cz, sz = np.cos(rx), np.sin(rx), np.cos(ry), np.sin(ry), np.cos(rz), np.sin(rz)
    Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]]); Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    return Rz @ Ry @ Rx


def _texture(w, h, seed):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    img = 128 + 60 * np.sin(xx / 7.3 + 0.02 * yy) * np.cos(yy / 5.1) + 25 * np.sin((xx + 2 * yy) / 23.0) + rng.normal(0, 6, (h, w))
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)


def other_cases():
    """name -> dict(kind=..., inputs...) -- small, deterministic, no OpenCV needed to build them."""
    out = []
    rigs = (("toe_in", 320, 240, (0.004, -0.06, 0.003), (1.0, 0.03, -0.02), 0.0),
            ("wide", 400, 260, (-0.01, 0.09, -0.004), (1.0, -0.05, 0.08), 6.0),
            ("mirror", 300, 200, (0.002, 0.04, 0.001), (-1.0, 0.01, 0.02), -4.0))           # T.x < 0: the rig wass_stereo swaps
    for name, w, h, ang, T, dc in rigs:
        f = 0.95 * w
        K1 = np.array([[f, 0, w / 2 + dc], [0, f * 1.01, h / 2 - dc / 2], [0, 0, 1.0]])
        K2 = np.array([[f * 0.98, 0, w / 2 - dc], [0, f * 0.985, h / 2 + 1.5], [0, 0, 1.0]])
        t = np.array(T, float); t /= np.linalg.norm(t)
        out.append(dict(name=f"rectify_{name}", kind="rectify", w=w, h=h, K1=K1, K2=K2, R=_rot(*ang), T=t, left=_texture(w, h, 11), right=_texture(w, h, 12)))
    for name, w, h, Hm in (("persp", 320, 200, np.array([[1.02, 0.03, -4.0], [-0.01, 0.98, 3.0], [2e-5, -1e-5, 1.0]])),
                          ("shift", 200, 120, np.array([[1, 0, 7.0], [0, 1, -3.0], [0, 0, 1.0]])),
                          ("zoom", 260, 180, np.array([[1.3, 0.0, -30.0], [0.0, 1.3, -20.0], [0, 0, 1.0]]))):
        out.append(dict(name=f"warp_{name}", kind="warp", H=Hm, src=_texture(w, h, 21)))
    for name, w, h, dist in (("k1k2", 320, 240, [-0.21, 0.09, 0.0, 0.0]), ("5coef", 400, 300, [-0.28, 0.11, 1e-3, -7e-4, -0.02]),
                             ("8coef", 300, 220, [0.1, -0.05, 5e-4, 2e-4, 0.01, 0.02, -0.01, 0.003]), ("tall", 5000 // 32, 300, [-0.1, 0.02, 0, 0])):
        K = np.array([[0.9 * w, 0, w / 2 + 3.0], [0, 0.92 * w, h / 2 - 2.0], [0, 0, 1.0]])
        out.append(dict(name=f"undistort_{name}", kind="undistort", K=K, dist=np.array(dist, float), src=_texture(w, h, 31)))
    for name, w, h, clip, tiles in (("default", 320, 240, 2.0, 8), ("ragged", 333, 251, 3.5, 8), ("tiles4", 200, 160, 1.0, 4), ("flat", 160, 120, 2.0, 8)):
        src = np.full((h, w), 77, np.uint8) if name == "flat" else _texture(w, h, 41)
        out.append(dict(name=f"clahe_{name}", kind="clahe", clip=clip, tiles=tiles, src=src))
    for name, w, h, fx, fy in (("half", 320, 240, 0.5, 0.5), ("x_only", 300, 200, 1.5, 1.0), ("third", 333, 251, 0.3, 0.3), ("up", 120, 90, 2.0, 2.0)):
        out.append(dict(name=f"resize_u8_{name}", kind="resize_u8", fx=fx, fy=fy, src=_texture(w, h, 51)))
    rng = np.random.default_rng(7)
    for name, w, h, ow, oh in (("down", 300, 200, 200, 200), ("up", 150, 100, 300, 100), ("both", 200, 150, 133, 100)):
        f = (_texture(w, h, 61).astype(np.float32) / 4.0) * (rng.random((h, w)) > 0.15)
        out.append(dict(name=f"resize_f32_{name}", kind="resize_f32", ow=ow, oh=oh, src=f.astype(np.float32)))
    for name, w, h, maxsize, maxdiff in (("small", 200, 120, 40, 16), ("big", 260, 180, 400, 32)):
        yy, xx = np.mgrid[0:h, 0:w]
        d = (16 * (12 + xx / 9.0 + yy / 31.0)).astype(np.int16)                     # a smooth 1/16-pixel disparity ramp ...
        for _ in range(60):                                                         # ... with islands of other values, 1 .. ~150 px
            cy, cx, ry, rx = rng.integers(0, h), rng.integers(0, w), rng.integers(1, 7), rng.integers(1, 14)
            d[max(cy - ry, 0):cy + ry, max(cx - rx, 0):cx + rx] = int(rng.integers(0, 60)) * 16
        d[rng.random((h, w)) > 0.97] = 0
        out.append(dict(name=f"speckle_{name}", kind="speckle", new_val=0, max_size=maxsize, max_diff=maxdiff, src=d))
    for name, w, h, thr in (("a", 200, 150, 400), ("b", 260, 170, 2500)):
        d = (_texture(w, h, 81).astype(np.float32) / 8.0)
        d[h // 3:h // 3 + 6, :] = 0
        out.append(dict(name=f"component_{name}", kind="component", threshold=thr, src=d))
    return out


def run_other(cv2, c):
    """The real library's answer for one case of other_cases(): dict of arrays."""
    k = c["kind"]
    if k == "rectify":
        size = (c["w"], c["h"])
        z = np.zeros(5)
        R1, R2, P1, P2, Q, roi1, roi2 = cv2.stereoRectify(c["K1"], z, c["K2"], z, size, c["R"], c["T"].reshape(3, 1), flags=0, alpha=1.0, newImageSize=size)
        mx1, my1 = cv2.initUndistortRectifyMap(c["K1"], None, R1, P1, size, cv2.CV_32FC1)                     # wass_stereo.cpp:600-601
        mx2, my2 = cv2.initUndistortRectifyMap(c["K2"], None, R2, P2, size, cv2.CV_32FC1)
        return dict(R1=R1, R2=R2, P1=P1, P2=P2, roi1=np.array(roi1), roi2=np.array(roi2), mx1=mx1, my1=my1, mx2=mx2, my2=my2,
                    left_rect=cv2.remap(c["left"], mx1, my1, cv2.INTER_CUBIC), right_rect=cv2.remap(c["right"], mx2, my2, cv2.INTER_CUBIC))   # :603-604
    if k == "warp":
        h, w = c["src"].shape
        return dict(dst=cv2.warpPerspective(c["src"], c["H"], (w, h)))                                       # :515-516
    if k == "undistort":
        return dict(dst=cv2.undistort(c["src"], c["K"], c["dist"]))                                          # wass_prepare.cpp:268
    if k == "clahe":
        return dict(dst=cv2.createCLAHE(clipLimit=c["clip"], tileGridSize=(c["tiles"], c["tiles"])).apply(c["src"]))   # wass_prepare.cpp:36-39,257-262
    if k == "resize_u8":
        return dict(dst=cv2.resize(c["src"], None, fx=c["fx"], fy=c["fy"], interpolation=cv2.INTER_CUBIC))   # wass_stereo.cpp:788-796, 413
    if k == "resize_f32":
        return dict(nearest=cv2.resize(c["src"], (c["ow"], c["oh"]), interpolation=cv2.INTER_NEAREST),        # :903-904
                    cubic=cv2.resize(c["src"], (c["ow"], c["oh"]), interpolation=cv2.INTER_CUBIC))
    if k == "speckle":
        d = c["src"].copy()
        cv2.filterSpeckles(d, c["new_val"], c["max_size"], c["max_diff"])
        return dict(dst=d)
    if k == "component":                                                                                      # :947-986
        d = c["src"]
        gx = cv2.Sobel(d, cv2.CV_32FC1, 1, 0); gy = cv2.Sobel(d, cv2.CV_32FC1, 0, 1)                        # :953-954 (3x3, BORDER_REFLECT_101)
        mag = gx * gx + gy * gy
        out = d.copy()
        out[mag > c["threshold"]] = 0.0                                                                       # :957-962
        n, labels, stats, _ = cv2.connectedComponentsWithStats((out != 0).astype(np.uint8) * 255)            # :964-970 (8-connectivity)
        best, max_area = 0, 0
        for i in range(1, n):                                                                                 # first strictly largest (:972-980)
            if stats[i, cv2.CC_STAT_AREA] > max_area:
                max_area, best = int(stats[i, cv2.CC_STAT_AREA]), i
        if best:
            out[labels != best] = 0.0
        return dict(sobel_mag=mag, dst=out)
    raise ValueError(k)


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden", "sgbm_opencv.npz"))
    ap.add_argument("--out-other", default=os.path.join(ROOT, "tests", "golden", "opencv_other.npz"),
                    help="rectification / undistort / CLAHE / resize / filterSpeckles / component vectors")
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
    other = {"opencv_version": np.array(ver)}
    onames = []
    for c in other_cases():
        for key, val in run_other(cv2, c).items():
            other[f"{c['name']}__{key}"] = np.asarray(val)
        onames.append(c["name"])
    other["names"] = np.array(onames)
    np.savez_compressed(args.out_other, **other)
    print(f"wrote {args.out_other}: {len(onames)} cases (rectify, warp, undistort, CLAHE, resize, filterSpeckles, component)")


if __name__ == "__main__":
    main()
