#!/usr/bin/env python3
"""scripts/host_cost.py [png_level] -- where the C++ sequence driver's host CPU per frame goes: one worker on 64 config-B workdirs in /dev/shm,
its own summary, WASS_PIPE_TIMING's submit() breakdown, one frame's time table, and the same run without the inlier text / with one writer."""
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                      # noqa: E402
from wass_amd import build        # noqa: E402

level = int(sys.argv[1]) if len(sys.argv) > 1 else 0
build.build_host()
tmp = tempfile.mkdtemp(prefix="wass_hostcost_", dir="/dev/shm")
try:
    seq, cfg, n = bench.make_sequence(tmp, 8, 8, 8, png_level=level)
    for extra, envx in (([], {}), ([], {"WASS_BLOCKING_SYNC": "0"}), (["--no-inliers-file"], {})):
        for d in os.listdir(seq):
            for f in ("mesh_cam.xyzC", "plane.txt"):
                try:
                    os.remove(os.path.join(seq, d, f))
                except OSError:
                    pass
        r = subprocess.run([build.BATCH, cfg, "--sequence", seq, "--gpus", "1"] + extra, capture_output=True, text=True, env=dict(os.environ, WASS_PIPE_TIMING="1", WASS_THREAD_CPU="1", **envx))
        print("==", " ".join(extra) or "default", envx, "rc", r.returncode)
        print("\n".join(l for l in r.stdout.splitlines() if "steady" in l or "host CPU" in l or "frame(s) ok" in l))
        print(r.stderr[-1800:])
    log = open(os.path.join(seq, "000020_wd", "wass_stereo_log.txt")).read()
    print(log[-1500:])
finally:
    shutil.rmtree(tmp, ignore_errors=True)
