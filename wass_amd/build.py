"""Build libwassgpu.so (HIP, gfx950 only) in-tree with hipcc.

    python -m wass_amd.build [--force]

The shared library lands in wass_amd/libwassgpu.so so that it travels with the
repo snapshot to the GPU box (it is git-ignored, not gpurun-ignored).
"""
from __future__ import annotations

import glob
import os
import subprocess
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
SO = os.path.join(_HERE, "libwassgpu.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# -ffp-contract=off: the fp64 geometry kernels must round like the reference's
# plain x86-64 build (no FMA contraction) to keep inlier counts bit-identical.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wall",
         "-Wno-unused-function", "-fno-gpu-rdc"] + os.environ.get("WASS_EXTRA_FLAGS", "").split()


def _newer(src: str, dst: str, deps) -> bool:
    if not os.path.exists(dst):
        return True
    t = os.path.getmtime(dst)
    return any(os.path.getmtime(p) > t for p in [src, *deps])


def build(force: bool = False, verbose: bool = False) -> str:
    os.makedirs(OBJ, exist_ok=True)
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    deps = sorted(glob.glob(os.path.join(CSRC, "*.h"))) + [os.path.join(_HERE, "..", "include", "wass_gpu.h")]
    # objects built with other flags (WASS_EXTRA_FLAGS experiments) must not be mixed with fresh ones: the checkpoint
    # distance, for one, is a compile-time constant shared by the cost stage and the aggregation
    stamp = os.path.join(OBJ, ".flags")
    try:
        hipcc_version = subprocess.run([HIPCC, "--version"], capture_output=True, text=True).stdout.strip().replace("\n", " | ")
    except OSError:
        hipcc_version = "?"
    flags_now = " ".join(FLAGS) + " || " + hipcc_version
    try:
        same = open(stamp).read() == flags_now
    except OSError:
        same = False
    if not same:
        # the stamp is only rewritten after a successful LINK: a build that dies between compiling and linking must not
        # leave a stamp that matches while the old library (other flags) is still in place
        force = True
        for stale in (stamp, SO):
            try:
                os.remove(stale)
            except OSError:
                pass
    objs, procs = [], []
    for s in srcs:
        o = os.path.join(OBJ, os.path.basename(s)[:-4] + ".o")
        objs.append(o)
        if force or _newer(s, o, deps):
            cmd = [HIPCC, *FLAGS, "-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd))
            procs.append((s, subprocess.Popen(cmd)))
    failed = [s for s, p in procs if p.wait() != 0]
    if failed:
        raise RuntimeError("hipcc failed for: " + ", ".join(failed))
    if force or procs or not os.path.exists(SO):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", SO, *objs, "-ldl"]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    with open(stamp, "w") as f:
        f.write(flags_now)
    return SO


HOST_DIR = os.path.join(_HERE, "host")
BIN_DIR = os.path.join(_HERE, "bin")
CLI = os.path.join(BIN_DIR, "wass_stereo")            # what wasscli starts: the client of the resident worker (no HIP behind it)
CLI_GPU = os.path.join(BIN_DIR, "wass_stereo_gpu")    # the full program: the client execs it for everything a server does not take
BATCH = os.path.join(BIN_DIR, "wass_stereo_batch")
PREPARE = os.path.join(BIN_DIR, "wass_prepare")


def build_host(force: bool = False, verbose: bool = False) -> str:
    """The drop-in wass_stereo / wass_prepare executables and the batch driver: plain C++17 (g++) above the C ABI, linked against libwassgpu.so."""
    build(force=False)
    os.makedirs(BIN_DIR, exist_ok=True)
    deps = sorted(glob.glob(os.path.join(HOST_DIR, "*.hpp"))) + [os.path.join(_HERE, "..", "include", "wass_gpu.h"), SO]
    for name, exe in (("wass_stereo_client.cpp", CLI), ("wass_stereo.cpp", CLI_GPU), ("wass_stereo_batch.cpp", BATCH), ("wass_prepare.cpp", PREPARE)):
        src = os.path.join(HOST_DIR, name)
        if force or _newer(src, exe, deps):
            cmd = [os.environ.get("CXX", "g++"), "-O2", "-std=c++17", "-Wall", "-Wno-unused-function", src, "-o", exe]
            if exe != CLI:                       # the per-frame client links nothing of ours: it must start in a millisecond
                cmd += ["-L" + _HERE, "-lwassgpu", "-lz", "-pthread", "-Wl,-rpath,$ORIGIN/..", "-Wl,-rpath-link," + "/opt/rocm/lib"]
            if verbose:
                print(" ".join(cmd))
            subprocess.check_call(cmd)
    return CLI


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
    print(build_host(force="--force" in sys.argv, verbose=True))
