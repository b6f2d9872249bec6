#!/usr/bin/env python3
"""scripts/server_env_ab.py [reps] -- the unchanged command line through the resident worker under alternating server environments (one prepared
config-B sequence, a fresh server per run): 4 concurrent callers (wasscli) and one caller at a time (matlab/run_wass.m)."""
import os, sys, time, subprocess, tempfile, shutil
sys.path.insert(0, os.getcwd())
from concurrent.futures import ThreadPoolExecutor
import bench
from wass_amd import build
build.build_host()
ENVS = [e for e in os.environ.get("AB_ENVS", "|WASS_SERVER_RA_THREADS=4|WASS_SERVER_RA_THREADS=4,WASS_SERVER_SPECULATE=5").split("|")]
tmp = tempfile.mkdtemp(prefix="wass_envab_", dir="/dev/shm")
try:
    seq, cfg, n = bench.make_sequence(tmp, 8, 12, 8)
    run = 0
    for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 2):
        for spec in ENVS:
            run += 1
            sock = os.path.join(tmp, "sock%d" % run); os.makedirs(sock)
            env = dict(os.environ, WASS_DEBUG_IMAGES="0", WASS_SERVER_DIR=sock, WASS_SERVER_IDLE="2")
            env.pop("WASS_NO_SERVER", None)
            env.update(dict(kv.split("=", 1) for kv in spec.split(",") if kv))
            def one(i, e=env):
                t = time.perf_counter()
                r = subprocess.run([build.CLI, cfg, os.path.join(seq, "%06d_wd" % i)], capture_output=True, env=e)
                return r.returncode, time.perf_counter() - t
            one(0)
            with ThreadPoolExecutor(4) as ex:
                t1 = time.perf_counter(); res = list(ex.map(one, range(1, n))); t2 = time.perf_counter()
            t3 = time.perf_counter(); seq1 = [one(i) for i in range(1, 41)]; t4 = time.perf_counter()
            envd = dict(env, WASS_DEBUG_IMAGES="1")
            one(0, envd)
            with ThreadPoolExecutor(4) as ex:
                t5 = time.perf_counter(); resd = list(ex.map(lambda i: one(i, envd), range(1, 41))); t6 = time.perf_counter()
            print("%-60s 4 callers %6.1f pairs/s | 1 caller %5.1f | 4 callers + debug pictures %5.1f | failed %d" % (spec or "(default)", (n - 1) / (t2 - t1), 40 / (t4 - t3), 40 / (t6 - t5),
                  sum(1 for rc, _ in res + seq1 + resd if rc)), flush=True)
finally:
    time.sleep(3)
    shutil.rmtree(tmp, ignore_errors=True)
