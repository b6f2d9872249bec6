#!/usr/bin/env python3
"""The shipped C++ sequence driver on ONE prepared config-B sequence, alternating environments (the sequence is built once):

    python scripts/cxx_ab.py [--rounds 3] [--replicate 16] [--prof DIR] "VAR=a" "VAR=b +driver-option=value" ...

Each run deletes the previous run's outputs first.  --prof DIR: additionally one rocprofv3 --kernel-trace --stats run of the first
environment on a short sequence, summary of the tail kernels printed."""
import argparse
import glob
import os
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from wass_amd import build  # noqa: E402

OUTPUTS = ("mesh_cam.xyzC", "plane.txt", "plane_refinement_inliers.xyz", "00000000_s.png", "00000001_s.png", "wass_stereo_log.txt")


def clean(seq):
    for wd in glob.glob(os.path.join(seq, "*_wd")):
        for f in OUTPUTS:
            try:
                os.remove(os.path.join(wd, f))
            except OSError:
                pass


def run(seq, cfg, env, extra=()):
    clean(seq)
    e = dict(os.environ)
    args = []
    for kv in env.split():
        if kv.startswith("+"):                                      # a driver option: +no-inliers-file, +decode-threads=12 ... (-- would be argparse's)
            args += ("--" + kv[1:]).split("=", 1)
            continue
        k, v = kv.split("=", 1)
        e[k] = v
    t0 = time.perf_counter()
    r = subprocess.run([*extra, build.BATCH, cfg, "--sequence", seq, "--gpus", "1", *args], capture_output=True, text=True, env=e)
    wall = time.perf_counter() - t0
    steady = cpu = None
    for line in r.stdout.splitlines():
        if line.startswith("steady state"):
            steady = float(line.split(":")[1].split()[0])
        if line.startswith("host CPU of the workers"):
            cpu = float(line.split(",")[1].split()[0])
    return r.returncode, steady, cpu, wall


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--replicate", type=int, default=16)
    ap.add_argument("--prof", default=None)
    ap.add_argument("envs", nargs="+")
    a = ap.parse_args()
    build.build_host()
    tmp = tempfile.mkdtemp(prefix="wass_cxx_ab_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    try:
        seq, cfg, n = bench.make_sequence(tmp, 8, a.replicate, 8)
        run(seq, cfg, a.envs[0])                                   # page cache, first-touch
        for i in range(a.rounds):
            for env in a.envs:
                rc, steady, cpu, wall = run(seq, cfg, env)
                print(f"{env:40s} rc={rc} steady {steady} pairs/s, {n / wall:.1f} incl. start-up, host CPU {cpu} ms/frame", flush=True)
        if a.prof:
            os.makedirs(a.prof, exist_ok=True)
            for wd in sorted(glob.glob(os.path.join(seq, "*_wd")))[24:]:
                shutil.rmtree(wd)
            os.environ["TMPDIR"] = "/tmp"
            rc, *_ = run(seq, cfg, a.envs[0] + " +in-process", extra=("rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", a.prof, "-o", "run", "--"))
            st = glob.glob(os.path.join(a.prof, "**", "*kernel_stats.csv"), recursive=True)
            if st:
                import csv
                print("kernel, calls, avg_us")
                for r in csv.DictReader(open(st[0])):
                    nm = r["Name"].split("(")[0].replace("void wass::", "").replace("wass::", "")
                    if float(r["TotalDurationNs"]) > 2e5:
                        print(f"  {nm[:50]:50s} {r['Calls']:>5s} {float(r['AverageNs']) / 1e3:9.1f}")
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
