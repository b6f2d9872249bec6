#!/usr/bin/env python3
"""Soak test of the shipped sequence driver: a long sequence through ONE wass_stereo_batch worker; every replica of a frame must
produce the same bytes as its original, and the worker's peak memory must not grow with the length of the sequence.

    python scripts/soak.py [--frames 2000] [--config A] [--tmp /dev/shm]
"""
import argparse, hashlib, os, resource, shutil, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=2000)
ap.add_argument("--distinct", type=int, default=8)
ap.add_argument("--config", default="A")
ap.add_argument("--tmp", default=None)
args = ap.parse_args()
import numpy as np  # noqa: E402
from test_cli import _write_xml  # noqa: E402
from wass_amd import build, synth  # noqa: E402
import bench  # noqa: E402
sys.path.insert(0, os.path.join(ROOT, 'scripts'))
from cli_throughput_png import write_png  # noqa: E402

w, h, D = bench.CONFIGS[args.config]
build.build_host()
tmp = tempfile.mkdtemp(prefix="wass_soak_", dir=args.tmp)
rig = synth.rig_geometry(w, h)
cfg = os.path.join(tmp, "stereo_config.txt")
open(cfg, "w").write(f"MAX_DISPARITY={D}\nRANDOM_SEED=12345\nUSE_CUSTOM_STEREORECTIFY=true\nRECTIFY_ANGLE=1e-6\nDISABLE_RECTIFY_ROI=true\n")


def make_sequence(seq, n):
    for i in range(min(n, args.distinct)):
        wd = os.path.join(seq, "%06d_wd" % i)
        os.makedirs(os.path.join(wd, "undistorted"))
        right, left = [t.numpy() for t in synth.make_pair_torch(w, h, D, frame_idx=i)]
        write_png(os.path.join(wd, "undistorted", "00000000.png"), left)
        write_png(os.path.join(wd, "undistorted", "00000001.png"), right)
        _write_xml(os.path.join(wd, "intrinsics_00000000.xml"), "intr", rig["K_left"])
        _write_xml(os.path.join(wd, "intrinsics_00000001.xml"), "intr", rig["K_right"])
        _write_xml(os.path.join(wd, "ext_R.xml"), "R", rig["R"])
        _write_xml(os.path.join(wd, "ext_T.xml"), "T", np.array(rig["T"]).reshape(3, 1) * 2.5)
    for i in range(args.distinct, n):
        src = os.path.join(seq, "%06d_wd" % (i % args.distinct)); dst = os.path.join(seq, "%06d_wd" % i)
        os.makedirs(os.path.join(dst, "undistorted"))
        for f in ("undistorted/00000000.png", "undistorted/00000001.png", "intrinsics_00000000.xml", "intrinsics_00000001.xml", "ext_R.xml", "ext_T.xml"):
            os.symlink(os.path.join(src, f), os.path.join(dst, f))


def run(n):
    seq = os.path.join(tmp, "seq%d" % n)
    make_sequence(seq, n)
    t0 = time.perf_counter()
    r = subprocess.run([build.BATCH, cfg, "--sequence", seq], capture_output=True, text=True)
    dt = time.perf_counter() - t0
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    rss = resource.getrusage(resource.RUSAGE_CHILDREN).ru_maxrss        # KiB, the largest of the driver and its workers so far
    worker = [l for l in r.stdout.splitlines() if "host CPU" in l or "frame(s) ok" in l]
    # every replica equals its original, byte for byte
    ref = {}
    bad = 0
    for i in range(n):
        wd = os.path.join(seq, "%06d_wd" % i)
        hs = tuple(hashlib.sha1(open(os.path.join(wd, f), "rb").read()).hexdigest() for f in ("mesh_cam.xyzC", "plane.txt", "plane_refinement_inliers.xyz"))
        k = i % args.distinct
        if k not in ref: ref[k] = hs
        elif ref[k] != hs: bad += 1
    nplanes = sum(1 for _ in open(os.path.join(seq, "planes.txt")))
    print(f"{n} frames: {dt:.1f} s = {n / dt:.1f} frames/s, largest process (driver / worker) max RSS so far {rss / 1024:.0f} MiB, planes.txt {nplanes} lines, replicas differing from their original: {bad}")
    for l in worker[-2:]: print("   ", l.strip())
    shutil.rmtree(seq, ignore_errors=True)
    return bad, nplanes


try:
    b1, p1 = run(max(args.distinct * 8, args.frames // 8))
    b2, p2 = run(args.frames)
    print("soak:", "PASSED" if b1 == 0 and b2 == 0 and p2 == args.frames else "FAILED")
finally:
    shutil.rmtree(tmp, ignore_errors=True)
