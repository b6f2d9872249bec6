"""debug helper: wass_stereo stage-by-stage vs pipelined on one synthetic workdir; prints the log lines that carry numbers"""
import os, shutil, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_cli import make_workdir
from wass_amd import build
cli = build.build_host()
extra = sys.argv[1].replace("\\n", "\n") if len(sys.argv) > 1 else ""
tmp = tempfile.mkdtemp()
wd, cfg, *_ = make_workdir(tmp, 400, 300, 64, extra_cfg=extra)
wd2 = os.path.join(tmp, "p_wd"); shutil.copytree(wd, wd2)
a = subprocess.run([cli, cfg, wd], capture_output=True, text=True, env=dict(os.environ, WASS_DEBUG_IMAGES="1", WASS_PIPE_DUMP=wd))
b = subprocess.run([cli, cfg, wd2], capture_output=True, text=True, env=dict(os.environ, WASS_DEBUG_IMAGES="0", WASS_PIPE_DUMP=wd2))
keep = ("valid points found", "biggest component size", "ransac rounds", "ransac plane coeffs", "refinement inliers", "estimated plane coeffs",
        "number of points after plane cropping", "total data size", "rectification map generated", "error")
for tag, r in (("sync", a), ("pipe", b)):
    print(tag, "rc", r.returncode)
    for l in r.stdout.splitlines():
        if any(k in l for k in keep):
            print("   ", l)
for name in ("mesh_cam.xyzC", "plane.txt", "plane_refinement_inliers.xyz", "P0cam.txt"):
    x, y = open(os.path.join(wd, name), "rb").read(), open(os.path.join(wd2, name), "rb").read()
    print(name, "equal" if x == y else f"DIFFER ({len(x)} vs {len(y)} bytes)")

import numpy as np
for name, dt in (("left_crop.bin", np.uint8), ("right_crop.bin", np.uint8), ("disp16.bin", np.int16), ("dispf.bin", np.float32)):
    x = np.fromfile(os.path.join(wd, name), dt); y = np.fromfile(os.path.join(wd2, name), dt)
    print(name, x.size, y.size, "equal" if x.size == y.size and np.array_equal(x, y) else ("DIFFER at %d elements, first %s" % ((x != y).sum(), np.flatnonzero(x != y)[:8]) if x.size == y.size else "SIZE"))
