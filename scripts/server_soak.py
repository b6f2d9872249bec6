#!/usr/bin/env python3
"""scripts/server_soak.py [calls] -- the resident worker under a long run of unchanged `wass_stereo <config> <workdir>` calls (4 at a time, half of
them with the reference's debug pictures): every call succeeds, replicas of a frame give byte-identical files, and the server's resident set
stops growing once its buffers exist (read from /proc/<pid>/status every 50 calls)."""
import hashlib
import os
import shutil
import subprocess
import sys
import tempfile
import time
from concurrent.futures import ThreadPoolExecutor

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
from wass_amd import build

calls = int(sys.argv[1]) if len(sys.argv) > 1 else 400
build.build_host()
base = "/dev/shm" if os.path.isdir("/dev/shm") else tempfile.gettempdir()
tmp = tempfile.mkdtemp(prefix="wass_soak_", dir=base)
sock = os.path.join(tmp, "sock")
os.makedirs(sock)
try:
    seq, cfg, n = bench.make_sequence(tmp, 8, 4, 8)                      # 32 workdirs, 8 distinct frames
    env = dict(os.environ, WASS_SERVER_DIR=sock, WASS_SERVER_IDLE="10")
    env.pop("WASS_NO_SERVER", None)

    def one(i):
        e = dict(env, WASS_DEBUG_IMAGES="1" if (i // n) % 2 else "0")
        r = subprocess.run([build.CLI, cfg, os.path.join(seq, "%06d_wd" % (i % n))], capture_output=True, text=True, env=e)
        return r.returncode

    def server_rss():
        out = subprocess.run(["ps", "-ww", "-eo", "pid,rss,args"], capture_output=True, text=True).stdout
        for l in out.splitlines():
            if "--server" in l and sock in l:
                return int(l.split()[1]) // 1024
        return -1

    def digest(i):
        h = hashlib.sha256()
        for f in ("mesh_cam.xyzC", "plane.txt", "plane_refinement_inliers.xyz", "stereo.jpg", "undistorted/R1.jpg", "graph_components.jpg"):
            p = os.path.join(seq, "%06d_wd" % i, f)
            h.update(open(p, "rb").read() if os.path.exists(p) else b"-")
        return h.hexdigest()

    rss, bad, t0 = [], 0, time.time()
    with ThreadPoolExecutor(4) as ex:
        for c0 in range(0, calls, 50):
            bad += sum(1 for rc in ex.map(one, range(c0, min(calls, c0 + 50))) if rc != 0)
            rss.append(server_rss())
    dt = time.time() - t0
    same = all(len({digest(i) for i in range(k, n, 8)}) == 1 for k in range(8))
    print(f"{calls} calls in {dt:.1f} s ({calls / dt:.1f}/s), failed {bad}, replicas identical: {same}")
    print("server RSS (MB) every 50 calls:", rss)
    third = max(1, len(rss) // 3)
    grow = max(rss[-third:]) - max(rss[1:1 + third] or rss[:1])          # (the first sample is taken while buffers are still being created)
    print("highest of the last third minus highest of the first third:", grow, "MB")
    sys.exit(0 if bad == 0 and same and grow < 64 else 1)
finally:
    time.sleep(0.3)
    shutil.rmtree(tmp, ignore_errors=True)
