#!/usr/bin/env python3
"""What wasscli would see: wall time per frame of the drop-in executables at config B (process start -> exit).

    python scripts/cli_throughput.py [--frames 8] [--config B]

Builds N synthetic workdirs (PNG + XML inputs as wass_prepare / wass_autocalibrate leave them), then times
  1. wass_stereo <cfg> <wd>              one process per frame, like wasscli (cli/wasscli/wasscli.py:326-346)
  2. wass_stereo_batch <cfg> --sequence   one worker process with a persistent context
"""
import argparse
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=8)
ap.add_argument("--config", default="B")
ap.add_argument("--procs", default="1,2,4", help="worker processes per GPU to try with wass_stereo_batch")
ap.add_argument("--threads", default="1", help="threads per worker process to try (each combined with every --procs value)")
ap.add_argument("--replicate", type=int, default=1, help="extra copies of every workdir (inputs symlinked) so that a long sequence is cheap to set up")
ap.add_argument("--skip-single", action="store_true", help="skip the one-process-per-frame runs")
ap.add_argument("--decode", default="6", help="decode threads per worker to try (pipelined chain)")
ap.add_argument("--writers", default="4", help="writer threads per worker to try")
ap.add_argument("--stage-by-stage", action="store_true", help="also time the synchronous per-stage workers")
ap.add_argument("--tmp", default=None, help="where the sequence lives (default: a fresh directory under the system's temp dir)")
args = ap.parse_args()
import numpy as np  # noqa: E402
from test_cli import _write_xml  # noqa: E402
from wass_amd import build, synth  # noqa: E402
import bench  # noqa: E402



def _write_png(path, img):                      # zlib level 1: the inputs only have to be valid PNG files, quickly
    import struct, zlib
    hh, ww = img.shape
    raw = b"".join(b"\x00" + img[y].tobytes() for y in range(hh))

    def chunk(t, d):
        c = struct.pack(">I", len(d)) + t + d
        return c + struct.pack(">I", zlib.crc32(t + d) & 0xFFFFFFFF)
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", ww, hh, 8, 0, 0, 0, 0)) +
                chunk(b"IDAT", zlib.compress(raw, 1)) + chunk(b"IEND", b""))


w, h, D = bench.CONFIGS[args.config]
cli = build.build_host()
tmp = tempfile.mkdtemp(prefix="wass_cli_", dir=args.tmp)
seq = os.path.join(tmp, "output")
rig = synth.rig_geometry(w, h)
cfg = os.path.join(tmp, "stereo_config.txt")
open(cfg, "w").write(f"MAX_DISPARITY={D}\nRANDOM_SEED=12345\nUSE_CUSTOM_STEREORECTIFY=true\nRECTIFY_ANGLE=1e-6\nDISABLE_RECTIFY_ROI=true\n")
for i in range(args.frames):
    wd = os.path.join(seq, "%06d_wd" % i)
    os.makedirs(os.path.join(wd, "undistorted"))
    right, left = [t.numpy() for t in synth.make_pair_torch(w, h, D, frame_idx=i)]
    _write_png(os.path.join(wd, "undistorted", "00000000.png"), left)
    _write_png(os.path.join(wd, "undistorted", "00000001.png"), right)
    _write_xml(os.path.join(wd, "intrinsics_00000000.xml"), "intr", rig["K_left"])
    _write_xml(os.path.join(wd, "intrinsics_00000001.xml"), "intr", rig["K_right"])
    _write_xml(os.path.join(wd, "ext_R.xml"), "R", rig["R"])
    _write_xml(os.path.join(wd, "ext_T.xml"), "T", np.array(rig["T"]).reshape(3, 1) * 2.5)


nframes = args.frames
for r in range(1, args.replicate):
    for i in range(args.frames):
        src = os.path.join(seq, "%06d_wd" % i)
        dst = os.path.join(seq, "%06d_wd" % nframes)
        os.makedirs(os.path.join(dst, "undistorted"))
        for f in ("undistorted/00000000.png", "undistorted/00000001.png", "intrinsics_00000000.xml", "intrinsics_00000001.xml", "ext_R.xml", "ext_T.xml"):
            os.symlink(os.path.join(src, f), os.path.join(dst, f))
        nframes += 1


def timed(cmd, env=None):
    t0 = time.perf_counter()
    r = subprocess.run(cmd, capture_output=True, text=True, env=env)
    return time.perf_counter() - t0, r


for dbg in (() if args.skip_single else ("1", "0")):
    env = dict(os.environ, WASS_DEBUG_IMAGES=dbg)
    ts = []
    for i in range(min(3, args.frames)):
        t, r = timed([cli, cfg, os.path.join(seq, "%06d_wd" % i)], env)
        assert r.returncode == 0, r.stdout[-2000:]
        ts.append(t)
    print(f"wass_stereo, one process per frame, debug pictures {'on' if dbg == '1' else 'off'}: {min(ts):.2f} s/frame (best of {len(ts)})")
    if dbg == "0":
        print("  time table of the last run:\n" + "\n".join(l for l in r.stdout.splitlines() if "|" in l and "P|" not in l))
def clean():
    for i in range(nframes):
        for f in ("mesh_cam.xyzC", "plane.txt", "plane_refinement_inliers.xyz"):
            try:
                os.remove(os.path.join(seq, "%06d_wd" % i, f))
            except OSError:
                pass


for procs in [int(x) for x in args.procs.split(",")]:
    for dec in [int(x) for x in args.decode.split(",")]:
        for wr in [int(x) for x in args.writers.split(",")]:
            for extra in ((), ("--no-inliers-file",)):
                clean()
                t, r = timed([build.BATCH, cfg, "--sequence", seq, "--procs-per-gpu", str(procs), "--decode-threads", str(dec), "--writer-threads", str(wr), *extra])
                assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
                print(f"wass_stereo_batch pipelined, {procs} worker(s), {dec} decode + {wr} writer threads{' ' + extra[0] if extra else ''}, {nframes} frames: "
                      f"{t:.2f} s total = {nframes / t:.2f} frames/s   [{r.stdout.strip().splitlines()[-1]}]")
if args.stage_by_stage:
    for procs, thr in [(int(x), int(y)) for x in args.procs.split(",") for y in args.threads.split(",")]:
        clean()
        t, r = timed([build.BATCH, cfg, "--sequence", seq, "--procs-per-gpu", str(procs), "--threads-per-proc", str(thr), "--stage-by-stage"])
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        print(f"wass_stereo_batch stage by stage, {procs} worker process(es) x {thr} thread(s) on one GPU, {nframes} frames: {t:.2f} s total = "
              f"{nframes / t:.2f} frames/s")
log = open(os.path.join(seq, "%06d_wd" % (nframes - 1), "wass_stereo_log.txt")).read()
print("  time table of the last frame of the last run:\n" + "\n".join(l for l in log.splitlines() if "|" in l and "P|" not in l))
import shutil  # noqa: E402
shutil.rmtree(tmp, ignore_errors=True)
