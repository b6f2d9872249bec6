#!/usr/bin/env python3
"""bench.py -- throughput of the wass_stereo dense-stereo hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--ndirs 5|8] [--config B|A|E]

One "step" = one synthetic rectified stereo pair pushed through the whole GPU hot path a1-a20 (SGBM, disparity
clean-up, triangulation, outlier removal, RANSAC plane + refinement, mesh_cam.xyzC image in host memory).  The timed region
cycles through up to 64 distinct frames per rank.  `value` is the rate of the pass that uploads both pictures of every frame
from pinned host memory inside the step, one frame ahead ("H2D of 2 x 5 MB images + all kernels + D2H of the
xyzC"; the boundary of the C++ driver hands over host buffers: decoded PNGs); the same steps with the inputs RESIDENT IN HBM
when the clock starts are timed in the same run.  Both passes are always reported under the stable names `pcie_inclusive` and
`resident_inputs` (rounds 1-3 and 5: value = pcie_inclusive; round 4: value = resident_inputs; `value_is` says which).  With N > 1
(one rank per GPU under torch.distributed.run; `--gpus N` without a launcher starts one) every rank processes its own
frames -- stereo frames are independent, there is no data-path collective -- and the reported value is the whole-job
rate (weak scaling); the only exchange is the 40-byte plane all-reduce after the timed region.

Prints ONE JSON line (rank 0) with BASELINE.json's metric plus
  roofline              -- path aggregation kernel family vs the 8 TB/s HBM roofline ((2R+4) B/cell)
  roofline_cost_volume  -- the cost-volume stage vs the packed-int16 VALU issue peak (it is not HBM-bound)
  cpu_baseline          -- the CPU oracle (5-path, whole path) timed on this box's host cores (rank 0, N = 1)
  cxx_driver            -- the shipped C++ sequence driver (wass_stereo_batch, one worker process per GPU) on a config-B sequence
                           of workdirs with every consumed output written: the product's own pairs/s, same run
  wasscli_unchanged     -- 4 concurrent `wass_stereo <config> <workdir>` processes (what wasscli starts, unedited) over a config-B
                           sequence, served by the per-GPU resident worker the first of them starts; `parallel_8`: the same with eight
                           (wasscli's menu setting); `server_ms_per_call`: where a call's time went inside the server
                           `one_caller_at_a_time`: a plain loop of calls (matlab/run_wass.m);
                           `with_debug_pictures`: the same with the reference's eight debug pictures per frame (its default), rendered and
                           JPEG-coded on the device
  mode_5path            -- config B in the mode the reference runs (MODE_SGBM, wass_stereo.cpp:775-777): pairs/s of the whole
                           chain, aggregation ms against its own (2*5+4) B/cell roofline
"""
from __future__ import annotations

import os as _os
# Six hardware queues instead of the runtime's four, set before PyTorch brings the HIP runtime up: every stream of the context (SGM, side,
# copy, tail) and the process's null stream then have a queue of their own.  With four, two of them share one and the runtime's choice
# among equally loaded queues differs from run to run -- tail + side on one queue costs the aggregation 0.35 ms, tail + SGM the whole
# tail (wass_amd/csrc/api.hip default_hw_queues; profiles/r06_x_streams.log).  The shipped executables set the same default themselves.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "6")
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CONFIGS = {
    # name: (w, h, D)   -- BASELINE.json configs[0], [1], [4]
    "A": (640, 480, 64),
    "B": (2456, 2058, 256),
    "E": (3840, 2160, 512),
}
HBM_PEAK_GBS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s
# Packed-u16 VALU peak: one wave instruction (64 lanes x 2 values) per SIMD every ~4.5 cycles, measured for every
# instruction kind of these kernels (v_pk_*_u16, v_min_u32, DPP moves) with scripts/micro/valu2.hip on MI355X;
# 1024 SIMDs x 2.4 GHz / 4.5 x 128 = 69.9 T ops/s.  (The guide quotes no integer-VALU figure.)
VALU_PK16_PEAK_TOPS = 1024 * 2.4e9 / 4.5 * 128 / 1e12
# arithmetic of one cost-volume cell, scalar view: per channel 4 saturating subtractions, 2 max,
# 1 min (x2 channels), raw >> 2, one add, +new -old for the horizontal and for the vertical sliding sum
COST_OPS_PER_CELL = 2 * 7 + 2 + 2 + 2


# ---------------------------------------------------------------------------------------------- CPU baseline
STAGES = ("Dense Stereo", "Triangulation", "Z-gap stats", "Outlier removal", "Plane fitting", "Plane refinement")


def _cpu_frame(job):
    """The reference's per-frame work (wass_stereo.cpp:1976-2127) with the CPU oracle, 5-path (= what the reference runs),
    one thread.  Stage names are the reference's timer events (:1977,1982,2047,2049,2065,2089)."""
    w, h, D, frame_idx = job
    sys.path.insert(0, ROOT)
    from oracle import oracle as O
    from wass_amd import synth
    if w * h <= 640 * 480:
        right, left = synth.make_pair(w, h, D, frame_idx=frame_idx)
    else:
        import torch
        torch.set_num_threads(1)
        right, left = [t.numpy() for t in synth.make_pair_torch(w, h, D, frame_idx)]
    rig = synth.rig_geometry(w, h)
    roi = (0, 0, w, h)
    mask = (right <= 254).astype(np.uint8)
    p = O.wass_params(D, mode=5)
    O.lib()
    t = [time.perf_counter()]
    d16, st = O.dense_disparity16(right, left, p)
    f = O.disparity_postprocess(d16, 1, D)
    t.append(time.perf_counter())
    n, valid, p3d, gray = O.triangulate(f, roi, roi, O.make_geom(rig), right, None, mask)
    t.append(time.perf_counter())
    zg, _ = O.zgap_percentile(valid, p3d, 99.0)
    t.append(time.perf_counter())
    valid, _ = O.keep_biggest_component(valid, p3d, zg)
    t.append(time.perf_counter())
    uv = O.ransac_sample(w, h, 400, 12345)
    ok, plane, best, _ = O.ransac_plane(valid, p3d, uv, 1.0)
    t.append(time.perf_counter())
    if ok:
        valid, _ = O.crop_plane(valid, p3d, plane, 1.0)
        plane, _, _ = O.refine_plane(valid, p3d)
        valid, _ = O.crop_plane(valid, p3d, plane, 1.5)
    blob = O.encode_xyzc(valid, p3d, plane)
    t.append(time.perf_counter())
    return {"stage_s": [t[i + 1] - t[i] for i in range(len(STAGES))], "total_s": t[-1] - t[0], "points": int(valid.sum()),
            "bytes": len(blob), "overflow": int(st.overflow)}


def _physical_cores():
    """distinct (physical id, core id) pairs of /proc/cpuinfo -- SMT siblings count once"""
    cores, phys, core = set(), None, None
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("physical id"):
                phys = line.split(":", 1)[1].strip()
            elif line.startswith("core id"):
                core = line.split(":", 1)[1].strip()
            elif not line.strip():
                if phys is not None and core is not None:
                    cores.add((phys, core))
                phys = core = None
    except OSError:
        pass
    return len(cores) or (os.cpu_count() or 1)


def _cpu_quota():
    """CPUs this container may actually burn: the cgroup CPU quota (cpu.max, v2 / cfs_quota_us, v1) and the affinity mask.  The GPU
    boxes of this pool show 256 hardware threads and grant 16 CPUs' worth of time: 128 processes measured 13.6x slower EACH than
    one alone (profiles/r03a_bench_driver_args_cpu128.json), i.e. they were time-sliced, not memory-bound."""
    q = float("inf")
    try:
        a, b = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if a != "max":
            q = float(a) / float(b)
    except (OSError, ValueError):
        try:
            a = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            b = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if a > 0:
                q = a / b
        except (OSError, ValueError):
            pass
    try:
        q = min(q, len(os.sched_getaffinity(0)))
    except AttributeError:
        pass
    return q


def _native_oracle():
    """The oracle's C sources compiled for THIS machine (-O3 -march=native, still without FMA contraction): the portable
    .so in oracle/ was built in another container.  Returns (path, flags) or (None, reason)."""
    import subprocess
    import tempfile
    src = [os.path.join(ROOT, "oracle", f) for f in ("sgbm_oracle.c", "wass_oracle.c", "rectify_oracle.c", "a9_oracle.c", "clahe_oracle.c")]
    flags = ["-O3", "-march=native", "-std=c99", "-fPIC", "-ffp-contract=off", "-fno-fast-math"]
    out = os.path.join(tempfile.gettempdir(), f"libwass_oracle_native_{os.getpid()}.so")
    try:
        subprocess.check_call([os.environ.get("CC", "gcc"), *flags, "-shared", "-o", out, *src, "-lm"], stderr=subprocess.DEVNULL)
        return out, " ".join(flags)
    except Exception as e:                                            # no compiler on the box: fall back to the portable build
        return None, f"native build failed ({type(e).__name__})"


def cpu_baseline(config: str):
    """SURVEY.md 8(d) / BASELINE.md section 3: the CPU restatement in 5-path mode (what the reference runs) on the SAME workload,
    one full frame per process, (i) on one thread and (ii) with one process per PHYSICAL host core at once, like wasscli's
    fan-out (cli/wasscli/wasscli.py:346); the oracle is compiled for this machine first (-O3 -march=native)."""
    import multiprocessing as mp
    w, h, D = CONFIGS[config]
    so, cflags = _native_oracle()
    if so:
        os.environ["WASS_ORACLE_LIB"] = so                             # inherited by the spawned workers
    one = _cpu_frame((w, h, D, 5000))
    threads = os.cpu_count() or 1
    phys, quota = _physical_cores(), _cpu_quota()
    n = max(1, int(min(phys, quota)))                             # one process per physical core the container may use
    ctx = mp.get_context("spawn")
    with ctx.Pool(n) as pool:
        pool.map(abs, range(n))                                   # processes up before the clock starts
        t0 = time.perf_counter()
        res = pool.map(_cpu_frame, [(w, h, D, 5001 + i) for i in range(n)], chunksize=1)
        wall = time.perf_counter() - t0
    if so:
        os.environ.pop("WASS_ORACLE_LIB", None)
        try:
            os.remove(so)
        except OSError:
            pass
    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    cells = w * h * D
    return {
        "value": round(n / wall, 4), "unit": "pairs/s", "cores": n, "kind": "port",
        "mdisp_per_sec": round(n / wall * cells / 1e6, 1),
        "sample": f"config {config} ({w}x{h}, D={D}), 5-path MODE_SGBM (what the reference runs), whole path a1-a20 with the scalar C "
                  f"oracle: one full frame per process, {n} processes at once = one per usable physical core ({phys} physical cores, "
                  f"{threads} hardware threads, cgroup CPU quota {quota:g}) "
                  f"({wall:.1f} s wall incl. input synthesis), after one full frame on one thread ({one['total_s']:.1f} s)",
        "physical_cores": phys, "hardware_threads": threads, "cpu_quota": (None if quota == float("inf") else quota),
        "cpu_model": model, "cflags": cflags,
        "single_thread": {"s_per_frame": round(one["total_s"], 2), "pairs_per_sec": round(1.0 / one["total_s"], 4),
                          "mdisp_per_sec": round(cells / one["total_s"] / 1e6, 1),
                          "stage_s": {k: round(v, 2) for k, v in zip(STAGES, one["stage_s"])}},
        "all_cores_stage_s_mean": {k: round(float(np.mean([r["stage_s"][i] for r in res])), 2) for i, k in enumerate(STAGES)},
        "note": "OpenCV's SGBM is SIMD-vectorised and would be faster than this scalar restatement by an unknown factor "
                "(est. 3-6x); the only published figure is ~30 s per 3 MP frame on a consumer i7 (doc/src/render/index.html.md:70)",
    }


def repeat_check(planes, npts, warmup, nf):
    """The timed region cycles through nf distinct frames: every revisit of a frame must give the SAME plane and point count,
    bit for bit (the pipeline overlaps uploads, the SGM stage and the tail of neighbouring frames on four streams; a missing
    ordering between them would show up here as a frame that changes with its neighbours)."""
    seen, revisits, bad = {}, 0, 0
    for i, (pl, n) in enumerate(zip(planes, npts)):
        k = (warmup + i) % nf
        key = (tuple(np.asarray(pl, np.float64).view(np.uint64).tolist()), int(n))
        if k in seen:
            revisits += 1
            bad += seen[k] != key
        else:
            seen[k] = key
    return {"frames_revisited": revisits, "mismatches": int(bad)}


def measured_traffic(config: str, ndirs: int):
    """Per-frame HBM bytes of the aggregation kernels from the committed rocprofv3 PMC summary of this command
    (profiles/*traffic_<config>_<ndirs>path.json, produced by scripts/profile.sh + scripts/traffic_json.py): the newest."""
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", f"*traffic_{config}_{ndirs}path.json"))):
        try:
            j = json.load(open(f))
            if j.get("config") == config and j.get("ndirs") == ndirs:
                cost = j.get("cost_stage_read_bytes_per_frame", 0) + j.get("cost_stage_write_bytes_per_frame", 0)
                best = (j["aggregation_hbm_bytes_per_frame"], os.path.relpath(f, ROOT), cost or None)
        except Exception:
            pass
    return best


def config_e_record(dev_index: int, ndirs: int, steps: int = 5, warmup: int = 2, config: str = "E"):
    """BASELINE.json configs[4] (3840x2160, D=512; the HBM-bound stress case) inside the default run: a few frames through the
    SGM stage a1-a6 with resident inputs, the same hipEvent brackets as the headline, so that the driver's BENCH record carries
    config E's aggregation time and roofline fraction next to config B's.  With config = "B": the headline configuration's SGM
    stage on its own, i.e. the aggregation kernels WITHOUT the previous frame's tail running underneath them."""
    import torch
    import wass_amd
    from wass_amd import synth
    w, h, D = CONFIGS[config]
    dev = torch.device("cuda", dev_index)
    params = wass_amd.default_sgm_params(D, ndirs=ndirs)
    frames = [synth.make_pair_torch(w, h, D, frame_idx=900000 + k, device=dev) for k in range(2)]
    out = torch.empty((h, w), dtype=torch.int16, device=dev)
    ctx = wass_amd.Context(dev_index)
    agg, tot, cost = [], [], []
    try:
        for i in range(warmup + steps):
            r, l = frames[i & 1]
            ctx.sgm_disparity_dev(r, l, params, out)
            if i >= warmup:
                ctx.synchronize()
                t = ctx.sgm_timings()
                agg.append(t.aggregate_ms); tot.append(t.total_ms); cost.append(t.cost_ms)
        ctx.synchronize()
        overflow = ctx.sgm_timings().cost_overflow
    finally:
        ctx.close()
    cells = w * h * D
    alg = cells * (2 * ndirs + 4)
    t_agg = float(np.mean(agg)) * 1e-3
    traffic = measured_traffic(config, ndirs)
    return {"workload": f"config {config}: {w}x{h}, D={D}, {ndirs}-path, SGM stage a1-a6, resident inputs, {steps} frames after {warmup}",
            "aggregate_ms": round(t_agg * 1e3, 3), "sgm_total_ms": round(float(np.mean(tot)), 3), "cost_volume_ms": round(float(np.mean(cost)), 3),
            "pairs_per_sec_sgm_stage": round(1e3 / float(np.mean(tot)), 2),
            "roofline": {"bound": "hbm", "achieved": round(alg / t_agg / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(alg / t_agg / 1e9 / HBM_PEAK_GBS, 4), "algorithmic_bytes": alg,
                         "traffic": int(traffic[0]) if traffic else None, "traffic_source": traffic[1] if traffic else None},
            "cost_overflow": int(overflow)}


def make_sequence(tmp: str, frames: int, replicate: int, ndirs: int, png_level: int = 1, raw: bool = False):
    """A config-B sequence of workdirs as wass_prepare / wass_autocalibrate leave them (PNG + XML) under tmp/output: `frames` distinct
    synthetic pairs, each workdir replicated `replicate` times with symlinked inputs.  Returns (sequence dir, config file, workdirs).
    png_level: 1 = deflated like cv::imwrite's default (the reference's wass_prepare); 0 = stored blocks (this product's wass_prepare).
    raw: instead of workdirs, what `wass_stereo_batch --raw` starts from -- tmp/calib (intrinsics, distortion, extrinsics) and
    tmp/input/cam{0,1}/<n>_frame.png, frames * replicate pictures per camera (symlinks to the distinct ones); returns (calib dir, config
    file, pictures per camera)."""
    import struct
    import zlib
    from wass_amd import synth
    w, h, D = CONFIGS["B"]

    def write_png(path, img):                    # zlib level 1: valid PNG files, quickly (the decoder's work is the same)
        raw = b"".join(b"\x00" + img[y].tobytes() for y in range(img.shape[0]))

        def chunk(t, d):
            c = struct.pack(">I", len(d)) + t + d
            return c + struct.pack(">I", zlib.crc32(t + d) & 0xFFFFFFFF)
        with open(path, "wb") as f:
            f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", img.shape[1], img.shape[0], 8, 0, 0, 0, 0)) +
                    chunk(b"IDAT", zlib.compress(raw, png_level)) + chunk(b"IEND", b""))

    def write_xml(path, node, m):
        m = np.atleast_2d(np.asarray(m, float))
        data = " ".join(repr(float(v)) for v in m.ravel())
        with open(path, "w") as f:
            f.write(f'<?xml version="1.0"?>\n<opencv_storage>\n<{node} type_id="opencv-matrix">\n  <rows>{m.shape[0]}</rows>\n'
                    f'  <cols>{m.shape[1]}</cols>\n  <dt>d</dt>\n  <data>\n    {data}</data></{node}>\n</opencv_storage>\n')

    seq = os.path.join(tmp, "output")
    rig = synth.rig_geometry(w, h)
    cfg = os.path.join(tmp, "stereo_config.txt")
    # WASS_BENCH_RECTIFY=opencv: the reference's DEFAULT rectification (cv::stereoRectify + remap, USE_CUSTOM_STEREORECTIFY=false) instead
    # of the built-in one with its ROI switched off -- for scripts/cli_unchanged.py; the bench line itself keeps the configuration below
    rect = "" if os.environ.get("WASS_BENCH_RECTIFY") == "opencv" else "USE_CUSTOM_STEREORECTIFY=true\nRECTIFY_ANGLE=1e-6\nDISABLE_RECTIFY_ROI=true\n"
    open(cfg, "w").write(f"MAX_DISPARITY={D}\nRANDOM_SEED=12345\n{rect}DENSE_PATHS={ndirs}\n")
    inputs = ("undistorted/00000000.png", "undistorted/00000001.png", "intrinsics_00000000.xml", "intrinsics_00000001.xml", "ext_R.xml", "ext_T.xml")
    if raw:
        calib = os.path.join(tmp, "calib")
        cams = [os.path.join(tmp, "input", "cam%d" % k) for k in (0, 1)]
        for d in [calib] + cams:
            os.makedirs(d)
        write_xml(os.path.join(calib, "intrinsics_00.xml"), "intr", rig["K_left"])
        write_xml(os.path.join(calib, "intrinsics_01.xml"), "intr", rig["K_right"])
        write_xml(os.path.join(calib, "distortion_00.xml"), "dist", np.array([-0.012, 0.004, 2e-4, -1e-4, 0.0]).reshape(5, 1))
        write_xml(os.path.join(calib, "distortion_01.xml"), "dist", np.array([0.009, -0.003, -1e-4, 2e-4, 1e-3]).reshape(5, 1))
        write_xml(os.path.join(calib, "ext_R.xml"), "R", rig["R"])
        write_xml(os.path.join(calib, "ext_T.xml"), "T", np.array(rig["T"]).reshape(3, 1) * 2.5)
        import torch
        dev = "cuda" if torch.cuda.is_available() else "cpu"
        for i in range(frames):
            right, left = [x.cpu().numpy() for x in synth.make_pair_torch(w, h, D, frame_idx=700000 + i, device=dev)]
            write_png(os.path.join(cams[0], "%06d_frame.png" % i), left)
            write_png(os.path.join(cams[1], "%06d_frame.png" % i), right)
        n = frames
        for _ in range(1, replicate):
            for i in range(frames):
                for c in cams:
                    os.symlink(os.path.join(c, "%06d_frame.png" % i), os.path.join(c, "%06d_frame.png" % n))
                n += 1
        return calib, cfg, n
    for i in range(frames):
        wd = os.path.join(seq, "%06d_wd" % i)
        os.makedirs(os.path.join(wd, "undistorted"))
        # (on the GPU when there is one: float64 sines over 5 megapixels take seconds on the host's cores, and torch starts a thread per
        # hardware thread it sees -- 256 on the pool's boxes, whose cgroup grants 16)
        try:
            import torch
            dev = "cuda" if torch.cuda.is_available() else "cpu"
            right, left = [t.cpu().numpy() for t in synth.make_pair_torch(w, h, D, frame_idx=700000 + i, device=dev)]
        except RuntimeError:
            right, left = [t.cpu().numpy() for t in synth.make_pair_torch(w, h, D, frame_idx=700000 + i)]
        write_png(os.path.join(wd, "undistorted", "00000000.png"), left)
        write_png(os.path.join(wd, "undistorted", "00000001.png"), right)
        write_xml(os.path.join(wd, "intrinsics_00000000.xml"), "intr", rig["K_left"])
        write_xml(os.path.join(wd, "intrinsics_00000001.xml"), "intr", rig["K_right"])
        write_xml(os.path.join(wd, "ext_R.xml"), "R", rig["R"])
        write_xml(os.path.join(wd, "ext_T.xml"), "T", np.array(rig["T"]).reshape(3, 1) * 2.5)
    n = frames
    for _ in range(1, replicate):
        for i in range(frames):
            dst = os.path.join(seq, "%06d_wd" % n)
            os.makedirs(os.path.join(dst, "undistorted"))
            for f in inputs:
                os.symlink(os.path.join(seq, "%06d_wd" % i, f), os.path.join(dst, f))
            n += 1
    return seq, cfg, n


def cxx_driver_record(ndirs: int, frames: int = 8, replicate: int = 24, decode_threads: int = 8, writer_threads: int = 4,
                      gpus: int = 1, procs_per_gpu: int = 1, png_level: int = 1, raw: bool = False):
    """What drops into wasscli: the C++ sequence driver (wass_amd/host/wass_stereo_batch.cpp, frame_pipeline.hpp -- decode threads
    -> device-resident frame chain -> writer threads) on a sequence of config-B workdirs as wass_prepare / wass_autocalibrate
    leave them (PNG + XML), one worker process on this GPU, every output a tool reads written (mesh_cam.xyzC, plane.txt, the
    camera files, the previews, the log; plane_refinement_inliers.xyz too).  `frames` distinct pairs, each workdir replicated
    `replicate` times with symlinked inputs.  The sequence lives in /dev/shm (memory-backed: 43 MB of output per frame).
    gpus > 1 (bench.py --gpus N, rank 0 after every rank has released its GPU): one worker process per GPU, `replicate` workdir
    copies PER GPU (fewer when the scratch directory cannot hold them) -- the product's own scaling point beside the harness's."""
    import shutil
    import struct
    import subprocess
    import tempfile
    import zlib
    from wass_amd import build, synth
    w, h, D = CONFIGS["B"]

    build.build_host()
    nworkers = max(1, gpus) * max(1, procs_per_gpu)
    base = None
    want = replicate * nworkers
    for rep in (want, max(replicate, want // 2), max(replicate // 2, want // 4), max(4, replicate // 2)):
        need = frames * rep * 45e6 + frames * 12e6 + 1e9          # outputs (43 MB per frame) + inputs + slack
        for cand in ("/dev/shm", tempfile.gettempdir()):
            try:
                if os.path.isdir(cand) and os.access(cand, os.W_OK) and shutil.disk_usage(cand).free > need:
                    base = cand
                    break
            except OSError:
                pass
        if base is not None:
            replicate = rep
            break
    if base is None:
        return {"error": f"no directory with {need / 1e9:.0f} GB free for the sequence (/dev/shm, {tempfile.gettempdir()})"}
    tmp = tempfile.mkdtemp(prefix="wass_bench_seq_", dir=base)
    try:
        seq, cfg, n = make_sequence(tmp, frames, replicate, ndirs, png_level=png_level, raw=raw)
        where = ["--sequence", seq]
        if raw:                                     # seq is the calibration directory here; the workdirs are created by the driver
            where = ["--raw", seq, "--cam0", os.path.join(tmp, "input", "cam0"), "--cam1", os.path.join(tmp, "input", "cam1"), "--sequence", os.path.join(tmp, "output")]
            seq = os.path.join(tmp, "output")
        t0 = time.perf_counter()
        r = subprocess.run([build.BATCH, cfg] + where + ["--gpus", str(max(1, gpus)), "--decode-threads", str(decode_threads),
                            "--writer-threads", str(writer_threads)] + (["--procs-per-gpu", str(procs_per_gpu)] if procs_per_gpu > 1 else []),
                           capture_output=True, text=True)
        wall = time.perf_counter() - t0
        if r.returncode != 0:
            return {"error": (r.stdout[-600:] + r.stderr[-600:]).strip(), "returncode": r.returncode}
        steady = cpu_ms = cores = None
        for line in r.stdout.splitlines():
            if line.startswith("steady state"):
                steady = float(line.split(":")[1].split()[0])
            if line.startswith("host CPU of the workers"):
                cores = float(line.split("(")[1].split()[0])
                cpu_ms = float(line.split(",")[1].split()[0])
        sizes = [os.path.getsize(os.path.join(seq, "%06d_wd" % (n - 1), f)) for f in ("mesh_cam.xyzC", "plane_refinement_inliers.xyz")]
        return {"pairs_per_sec": round(steady, 2) if steady else None, "pairs_per_sec_incl_startup": round(n / wall, 2), "seconds": round(wall, 2),
                "frames": n, "distinct_frames": frames, "workers": nworkers, "gpus": max(1, gpus), "decode_threads": decode_threads,
                "writer_threads": writer_threads,
                "ndirs": ndirs, "pipelined": "pipelined" in r.stdout, "host_cpu_ms_per_frame": cpu_ms, "host_cores_busy": cores,
                "inputs": ("the cameras' raw pictures (PNG, zlib level 1) + calibration directory: wass_stereo_batch --raw, undistortion on the device" if raw else
                           "workdirs as the reference's wass_prepare leaves them: undistorted/*.png deflated at zlib level 1 (cv::imwrite's default)" if png_level else
                           "workdirs as THIS product's wass_prepare leaves them: undistorted/*.png as stored blocks (valid PNG, inflate = copy)"),
                "outputs": "all files wass_stereo writes without its debug pictures, per workdir: mesh_cam.xyzC "
                           f"({sizes[0] / 1e6:.1f} MB), plane.txt, plane_refinement_inliers.xyz ({sizes[1] / 1e6:.1f} MB), camera / pose files, "
                           "scaled previews, stereo_config.txt, wass_stereo_log.txt; planes.txt + planes_mean.txt for the sequence",
                "note": "pairs_per_sec = (computed frames - 1) / (last finished - first finished) of the worker, i.e. without HIP start-up; "
                        "the other rate is the whole process from fork to exit",
                "where": base}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def wasscli_unchanged_record(ndirs: int, frames: int = 8, replicate: int = 12, parallel: int = 4, debug_images: bool = False):
    """What wasscli gets WITHOUT being edited: cli/wasscli/wasscli.py:326-346 starts one `wass_stereo <config> <workdir>` process per
    frame, NUM_PARALLEL_PROCESSES (4) at a time.  The same here -- `parallel` concurrent wass_stereo processes over a config-B sequence
    of workdirs -- with the executable handing its frame to the per-GPU resident worker it starts on demand (wass_amd/host/
    stereo_server.hpp), and, for comparison, a few frames with WASS_NO_SERVER=1 (every process initialises HIP and computes its own
    frame: round 4).  WASS_DEBUG_IMAGES=0 as in cxx_driver (the reference's eight debug JPEGs per frame are half a second of host time)."""
    import shutil
    import subprocess
    import tempfile
    from concurrent.futures import ThreadPoolExecutor
    from wass_amd import build
    build.build_host()
    base = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else tempfile.gettempdir()
    tmp = tempfile.mkdtemp(prefix="wass_bench_cli_", dir=base)
    sockdir = os.path.join(tmp, "sock")
    os.makedirs(sockdir)
    try:
        seq, cfg, n = make_sequence(tmp, frames, replicate, ndirs)
        tlog = os.path.join(tmp, "server_timing.log")
        env = dict(os.environ, WASS_DEBUG_IMAGES="1" if debug_images else "0", WASS_SERVER_DIR=sockdir, WASS_SERVER_IDLE="5", WASS_SERVER_TIMING=tlog)
        env.pop("WASS_NO_SERVER", None)

        def one(i, e):
            t = time.perf_counter()
            r = subprocess.run([build.CLI, cfg, os.path.join(seq, "%06d_wd" % i)], capture_output=True, text=True, env=e)
            return r.returncode, time.perf_counter() - t, time.perf_counter()
        # the first caller starts the server (HIP start-up, scratch allocation): reported separately
        t0 = time.perf_counter()
        rc0, first_s, _ = one(0, env)
        with ThreadPoolExecutor(parallel) as ex:
            t1 = time.perf_counter()
            res = list(ex.map(lambda i: one(i, env), range(1, n)))
            t2 = time.perf_counter()
        bad = [rc for rc, _, _ in res if rc != 0] + ([rc0] if rc0 != 0 else [])
        lat = sorted(s for _, s, _ in res)
        rec = {"pairs_per_sec": round((n - 1) / (t2 - t1), 2), "pairs_per_sec_incl_server_start": round(n / (t2 - t0), 2), "frames": n,
               "parallel_processes": parallel, "first_call_s": round(first_s, 3), "median_call_s": round(lat[len(lat) // 2], 4),
               "failed_calls": len(bad), "ndirs": ndirs,
               "how": f"{parallel} concurrent `wass_stereo <config> <workdir>` processes (a thread pool, as wasscli's thread_map) over {n} config-B "
                      "workdirs; each hands its frame to the per-GPU resident worker started by the first; WASS_DEBUG_IMAGES=" + ("1" if debug_images else "0") + "; output to " + base}
        # where a caller's waiting time went, as the server saw it (WASS_SERVER_TIMING: medians over the frames above)
        try:
            rows = [l.split() for l in open(tlog) if " total " in l][1:]
            med = lambda k: round(sorted(float(r[r.index(k) + 1]) for r in rows)[len(rows) // 2], 1)
            rec["server_ms_per_call"] = {k: med(k) for k in ("decode", "queue", "gpu", "files", "total")}
        except Exception as e:
            rec["server_ms_per_call"] = {"error": f"{type(e).__name__}: {e}"}
        # wasscli's own menu ("Set number of parallel workers", wasscli.py:440-454) raises NUM_PARALLEL_PROCESSES without an edit: 8 callers
        with ThreadPoolExecutor(2 * parallel) as ex:
            t1 = time.perf_counter()
            res8 = list(ex.map(lambda i: one(i, env), range(1, n)))
            t2 = time.perf_counter()
        rec["parallel_%d" % (2 * parallel)] = {"pairs_per_sec": round((n - 1) / (t2 - t1), 2), "failed_calls": len([1 for rc, _, _ in res8 if rc != 0]),
                                              "median_call_s": round(sorted(s for _, s, _ in res8)[len(res8) // 2], 4)}
        try:
            rows = [l.split() for l in open(tlog) if " total " in l][n + 1:]
            rec["parallel_%d" % (2 * parallel)]["server_ms_per_call"] = {k: med(k) for k in ("decode", "queue", "gpu", "files", "total")}
        except Exception:
            pass
        # one call after the other, what matlab/run_wass.m:242-246 and test/test_pipeline.m:141-143 do: the worker computes the next frames
        # while the caller is between two calls
        if parallel > 1:
            ns = min(n - 1, 40)
            t1 = time.perf_counter()
            res1 = [one(i, env) for i in range(1, ns + 1)]
            t2 = time.perf_counter()
            rec["one_caller_at_a_time"] = {"pairs_per_sec": round(ns / (t2 - t1), 2), "calls": ns, "failed_calls": len([1 for rc, _, _ in res1 if rc != 0]),
                                           "median_call_s": round(sorted(s for _, s, _ in res1)[ns // 2], 4)}
        # the reference's own default: its eight debug pictures per frame (cv::imwrite, unconditional) -- here rendered and JPEG-coded on the
        # device (csrc/jpeg.hip); the numbers above are with WASS_DEBUG_IMAGES=0
        if not debug_images:
            envd = dict(env, WASS_DEBUG_IMAGES="1")
            nd = min(n, 33)
            one(0, envd)                                                  # (the pipeline of this option set: buffers)
            with ThreadPoolExecutor(parallel) as ex:
                t1 = time.perf_counter()
                resd = list(ex.map(lambda i: one(i, envd), range(1, nd)))
                t2 = time.perf_counter()
            pics = [f for f in ("stereo.jpg", "stereo_input.jpg", "disparity_stereo_ouput.jpg", "disparity_final_scaled.jpg", "disparity_coverage.jpg",
                                "graph_components.jpg", "undistorted/R0.jpg", "undistorted/R1.jpg") if os.path.exists(os.path.join(seq, "%06d_wd" % 1, f))]
            rec["with_debug_pictures"] = {"pairs_per_sec": round((nd - 1) / (t2 - t1), 2), "frames": nd - 1, "failed_calls": len([1 for rc, _, _ in resd if rc != 0]),
                                          "median_call_s": round(sorted(s for _, s, _ in resd)[len(resd) // 2], 4), "pictures_per_frame": len(pics),
                                          "picture_bytes_per_frame": sum(os.path.getsize(os.path.join(seq, "%06d_wd" % 1, f)) for f in pics)}
        # round 4's behaviour on a few frames: every process computes its own frame
        env0 = dict(env, WASS_NO_SERVER="1")
        for i in range(8):
            for f in ("mesh_cam.xyzC", "plane.txt"):
                try:
                    os.remove(os.path.join(seq, "%06d_wd" % i, f))
                except OSError:
                    pass
        with ThreadPoolExecutor(parallel) as ex:
            t1 = time.perf_counter()
            res0 = list(ex.map(lambda i: one(i, env0), range(8)))
            t2 = time.perf_counter()
        rec["without_server"] = {"pairs_per_sec": round(8 / (t2 - t1), 2), "frames": 8, "failed_calls": len([1 for rc, _, _ in res0 if rc != 0]),
                                 "median_call_s": round(sorted(s for _, s, _ in res0)[4], 3)}
        return rec
    finally:
        time.sleep(0.2)
        shutil.rmtree(tmp, ignore_errors=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=128)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--ndirs", type=int, default=8, choices=(5, 8))
    ap.add_argument("--config", default="B", choices=sorted(CONFIGS))
    ap.add_argument("--frames", type=int, default=64, help="distinct frames per rank (cycled when steps exceed it)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-config-e", action="store_true", help="skip the short config E (3840x2160, D=512) sub-record of the default run")
    ap.add_argument("--no-tail-overlap", action="store_true",
                    help="run the post-SGM stages on the SGM stream instead of the context's tail stream")
    ap.add_argument("--uploads", action="store_true", help="(default since round 5; kept for old command lines)")
    ap.add_argument("--resident-inputs", action="store_true",
                    help="make the resident-input pass the main timed region (`value`); kernel-path number, not the contract's metric")
    ap.add_argument("--no-pcie-pass", "--no-second-pass", dest="no_pcie_pass", action="store_true",
                    help="skip the second pass (the one that is not `value`)")
    ap.add_argument("--no-5path", action="store_true", help="skip the 5-path (MODE_SGBM) sub-record of the default run")
    ap.add_argument("--inlier-text", action="store_true",
                    help="also produce plane_refinement_inliers.xyz (every 10th refinement inlier, text formatted on the device) per frame, like "
                         "the C++ driver does; not part of the metric's pass")
    ap.add_argument("--no-cxx-driver", action="store_true", help="skip the C++ sequence driver's own throughput (cxx_driver) and wasscli_unchanged")
    ap.add_argument("--stage", default="full", choices=("full", "sgm"),
                    help="full = a1-a20 (SGBM, clean-up, triangulation, plane fit, xyzC); sgm = a1-a6 only")
    ap.add_argument("--rccl-single-rank", action="store_true",
                    help="functional test: run the multi-rank epilogue (RCCL through the C ABI included) with a process group of one rank")
    ap.add_argument("--allow-shared-gpu", action="store_true",
                    help="let several ranks share one GPU (rank r uses device r %% device_count): only for exercising the "
                         "multi-rank path on a single-GPU box; throughput numbers are then meaningless")
    args = ap.parse_args()

    # --gpus N without a launcher: become the launcher (one rank per GPU under torch.distributed.run, rendezvous on
    # 127.0.0.1).  Under a launcher WORLD_SIZE must agree with --gpus: a silent 1-rank run of an N-GPU request is an error.
    if "WORLD_SIZE" not in os.environ:
        if args.gpus > 1:
            import socket
            import subprocess
            with socket.socket() as so:
                so.bind(("127.0.0.1", 0))
                port = so.getsockname()[1]
            cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
                   "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
            sys.exit(subprocess.call(cmd))
    elif int(os.environ["WORLD_SIZE"]) != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={os.environ['WORLD_SIZE']} ranks")

    import torch
    import wass_amd
    from wass_amd import synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    ndev = torch.cuda.device_count()
    if ndev == 0:
        sys.exit("bench.py: no GPU visible (libwassgpu has no CPU path)")
    if local_rank >= ndev and not args.allow_shared_gpu:
        sys.exit(f"bench.py: rank {rank} needs GPU {local_rank} but only {ndev} device(s) are visible "
                 f"(--allow-shared-gpu runs the ranks on shared devices, for functional tests only)")
    dev_index = local_rank % ndev
    torch.cuda.set_device(dev_index)
    if world == 1 and args.rccl_single_rank:
        # functional test of everything a multi-rank run does after the timed region -- process-group collectives, the library's
        # own RCCL all-reduce (wass_coll_*), the gathered per-rank rates -- with a process group of ONE rank on one GPU
        import socket
        import torch.distributed as dist
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(port))
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", dev_index))
    if world > 1:
        import torch.distributed as dist
        # RCCL refuses two ranks on one device; the functional single-GPU test of the multi-rank path uses gloo
        backend = "gloo" if (args.allow_shared_gpu and world > ndev) else "nccl"
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group("gloo")
        world = dist.get_world_size()                    # what the process group actually has, not what the env said
    dev = torch.device("cuda", dev_index)
    coll_dev = dev if (dist is None or dist.get_backend() == "nccl") else torch.device("cpu")

    w, h, D = CONFIGS[args.config]
    params = wass_amd.default_sgm_params(D, ndirs=args.ndirs)
    ctx = wass_amd.Context(dev_index)
    tail_overlap = args.stage == "full" and not args.no_tail_overlap
    ctx.set_tail_overlap(tail_overlap)

    # distinct frames per rank, synthesised on the GPU (wass_amd.synth.make_pair_torch == make_pair), parked in pinned
    # host memory: the timed region uploads them like a sequence driver would after decoding the PNGs
    # every frame of the timed region comes round at least twice (repeat_check), also at the driver's --steps 20
    nf = max(2, min(args.frames, max(2, args.steps // 2)))
    host = []
    for k in range(nf):
        r, l = synth.make_pair_torch(w, h, D, frame_idx=rank * 100000 + k, device=dev)
        host.append((r.cpu().pin_memory(), l.cpu().pin_memory()))
    del r, l
    torch.cuda.synchronize()
    NBUF = 6                                             # two frames staged ahead, the one being submitted, two pending, one whose tail may still read its picture (FramePipeline::NIN)
    # resident pass: every distinct frame (both pictures + the burned-area mask of the right one) sits in HBM before the clock
    # starts.  PCIe pass: a ring of six input sets, frame i+2 uploaded from pinned memory right before frame i is submitted
    # (what the C++ driver does after decoding: frame_pipeline.hpp stages two frames ahead) through wass_upload_async -- the context's
    # copy stream + an event the SGM stream waits for.  Two ahead, not one (round 6): the copy stream carries frame i's downloads, which
    # wait for its tail, in front of the next upload; one ahead, the SGM stage of frame i+2 had 0.4 ms of slack behind the tail of
    # frame i (NOTES/measurement.md "Round 6") -- a box whose tail kernels stretch paid that in pairs/s.
    dres = [tuple(torch.empty((h, w), dtype=torch.uint8, device=dev) for _ in range(3)) for _ in range(nf)]
    dring = [tuple(torch.empty((h, w), dtype=torch.uint8, device=dev) for _ in range(3)) for _ in range(NBUF)]
    for k in range(nf):
        dres[k][0].copy_(host[k][0]); dres[k][1].copy_(host[k][1])
        dres[k][2].copy_(dres[k][0] <= 254)
    torch.cuda.synchronize()
    geom = wass_amd.make_geom(synth.rig_geometry(w, h))
    # wass_stereo.cpp main() per frame: SGM -> clean-up -> triangulate -> z-gap / biggest component -> RANSAC -> crop ->
    # refine -> crop -> mesh_cam.xyzC (defaults of SURVEY.md Appendix C, RANDOM_SEED=12345), as wass_amd.batch.FramePipeline
    # enqueues it: no host synchronisation inside a frame, the previous frame's output is collected while this one runs
    from wass_amd.batch import FramePipeline
    pipe = FramePipeline(ctx, w, h, params, geom, tail_overlap=tail_overlap, inliers_text=args.inlier_text) if args.stage == "full" else None
    sgm_out = torch.empty((h, w), dtype=torch.int16, device=dev)

    def run_pass(resident: bool, steps: int, warmup: int, pipe=pipe, params=params, kernel_events: bool = False):
        """W untimed steps, then exactly `steps` timed ones between two barriers; returns what the JSON line needs.
        kernel_events (never in a pass whose rate is reported): every launch of the cost stage and of the aggregation family bracketed by
        hipEvents on its own stream, read after every step -- the SGM streams are waited for, the frame's tail still runs underneath the
        next frame's kernels."""
        planes, npts_hist, nbytes_hist, overflows = [], [], [], []
        kms = {}

        def keep(o):
            if o is not None:
                planes.append(o.plane); npts_hist.append(o.n_points); nbytes_hist.append(len(o.xyzc)); overflows.append(o.cost_overflow)

        def upload(i):
            dr, dl, dm = dring[i % NBUF]
            ctx.upload_async(dr, host[i % nf][0])
            ctx.upload_async(dl, host[i % nf][1])

        def step(i):
            if resident:
                dr, dl, dm = dres[i % nf]
            else:
                # one upload per step, of the frame after next: the transfer runs on the copy stream underneath frame i; buffer
                # (i+2) % 6 was last used by frame i-4
                dr, dl, dm = dring[i % NBUF]
                if i == 0:
                    upload(0)
                    upload(1)
                ctx.burned_area_mask_dev(dr, dm)            # DISCARD_BURNED_AREAS mask of the right image (wass_stereo.cpp:1072)
                upload(i + 2)
            if args.stage == "sgm":
                ctx.sgm_disparity_dev(dr, dl, params, sgm_out)
            else:
                keep(pipe.submit(dr, dl, d_right_image=dr, d_right_mask=dm))

        def barrier():
            if pipe is not None:
                for o in pipe.drain():
                    keep(o)
            torch.cuda.synchronize()
            if dist is not None:
                dist.barrier()
            torch.cuda.synchronize()

        for i in range(warmup):
            step(i)
        barrier()
        planes.clear(); npts_hist.clear(); nbytes_hist.clear(); overflows.clear()
        tm = {"agg": [], "cost": [], "sel": [], "sgm": [], "vsum": [], "pre": [], "med": []}

        def take(t):
            tm["agg"].append(t.aggregate_ms); tm["cost"].append(t.cost_ms); tm["sel"].append(t.select_ms); tm["sgm"].append(t.total_ms)
            tm["vsum"].append(t.vsum_ms); tm["pre"].append(t.prefilter_ms); tm["med"].append(t.median_ms)

        # Stage timings come from hipEvents recorded on the context's own stream; call n's are read after call n+2 has been
        # enqueued (the library keeps four sets), so the reader never waits for a frame that is still running
        if kernel_events:
            ctx.set_kernel_events(True)
        t0 = time.perf_counter()
        c0 = ctx.sgm_call_count()
        try:
            for i in range(steps):
                step(warmup + i)
                if kernel_events:
                    for name, ms in ctx.sgm_kernel_times():
                        kms.setdefault(name, []).append(ms)
                if i > 1:
                    take(ctx.sgm_call_timings(c0 + i - 1))
            barrier()
        finally:
            if kernel_events:
                ctx.set_kernel_events(False)
        for k in range(max(steps - 2, 0), steps):
            take(ctx.sgm_call_timings(c0 + k + 1))
        return {"elapsed": time.perf_counter() - t0, "planes": planes, "npts": npts_hist, "nbytes": nbytes_hist, "overflows": overflows, "tm": tm,
                "kernel_ms": {k: round(float(np.mean(v)), 3) for k, v in kms.items()}}

    # the metric's pass includes the H2D of both pictures.  (Round 4 had the resident pass as `value`; the two
    # differ by a fraction of a percent -- the 10 MB upload hides under the previous frame -- and both are always reported.)
    resident_main = bool(args.resident_inputs)
    main_pass = run_pass(resident_main, args.steps, args.warmup)
    elapsed = main_pass["elapsed"]
    planes, npts_hist, nbytes_hist, overflows = main_pass["planes"], main_pass["npts"], main_pass["nbytes"], main_pass["overflows"]
    agg_ms, cost_ms, sel_ms, sgm_ms, vsum_ms = (main_pass["tm"][k] for k in ("agg", "cost", "sel", "sgm", "vsum"))
    other_pass = None
    if world == 1 and not args.no_pcie_pass:
        other_pass = run_pass(not resident_main, args.steps, min(args.warmup, 5))
    # The mode the reference actually runs: StereoSGBM::create leaves MODE_SGBM, five paths (wass_stereo.cpp:775-777).  Same
    # frames, same chain, same brackets, inputs uploaded inside the step like the headline pass.
    # what the column paths add to the cost stage's vertical sum, measured on the last frame's horizontal sums (plain sum vs
    # the production kernel, best of three each, outside the timed region).  The probe re-launches the LAST call's form of the kernel,
    # so it comes right behind the pass it belongs to (round 5 probed after the 5-path pass and charged the 8-path record with the
    # 5-path kernel).
    vsum_probe = ctx.sgm_probe_vsum()
    # per-kernel times of the family (hipEvents around every launch, a few pipelined frames outside the timed passes): the record
    # can be re-derived from the line alone
    kernel_ms = None
    if world == 1 and args.stage == "full":
        kernel_ms = run_pass(resident_main, 6, 2, kernel_events=True)["kernel_ms"]
    pass5 = vsum_probe5 = kernel_ms5 = None
    if world == 1 and args.config == "B" and args.stage == "full" and args.ndirs == 8 and not args.no_5path:
        params5 = wass_amd.default_sgm_params(D, ndirs=5)
        pipe5 = FramePipeline(ctx, w, h, params5, geom, tail_overlap=tail_overlap)
        pass5 = run_pass(False, min(args.steps, 64), min(args.warmup, 5), pipe=pipe5, params=params5)
        vsum_probe5 = ctx.sgm_probe_vsum()
        kernel_ms5 = run_pass(False, 6, 2, pipe=pipe5, params=params5, kernel_events=True)["kernel_ms"]
    # Coll-1: sequence mean plane = NaN-aware mean over every rank's frames (5 doubles all-reduced over RCCL)
    acc = wass_amd.planes_mean_accumulate(np.array(planes).reshape(-1, 4)) if planes else np.zeros(5)
    rank_rates = [args.steps / elapsed]
    coll_info = None
    if dist is not None:
        acc_t = torch.tensor(acc, dtype=torch.float64, device=coll_dev)
        dist.all_reduce(acc_t, op=dist.ReduceOp.SUM)
        acc_torch = acc_t.cpu().numpy()
        if dist.get_backend() == "nccl":
            # Coll-1 as the product does it (wass_stereo_batch): ncclCommInitRank + ncclAllReduce(ncclSum, ncclDouble) through the
            # C ABI, on the context's own stream; the unique id travels over the process group.  The torch all-reduce above is
            # only the cross-check.
            import ctypes as C
            from wass_amd import _lib
            lib = _lib.load()
            uid_t = torch.zeros(128, dtype=torch.uint8, device=coll_dev)
            # A failure of this leg is REPORTED in the line (plane_allreduce.error), it does not take the throughput
            # measurement down with it: the timed region is over, and the mean then comes from the torch all-reduce.
            ok_t = torch.ones(1, dtype=torch.int32, device=coll_dev)
            if rank == 0:
                uid = (C.c_ubyte * 128)()
                if lib.wass_coll_unique_id(uid) != 0:
                    ok_t.zero_()
                else:
                    uid_t = torch.tensor(list(uid), dtype=torch.uint8, device=coll_dev)
            dist.broadcast(ok_t, 0)
            coll_err = None if int(ok_t.item()) else "wass_coll_unique_id failed on rank 0 (librccl not loadable)"
            if coll_err is None:
                dist.broadcast(uid_t, 0)
                uid = (C.c_ubyte * 128)(*uid_t.cpu().tolist())
                a5 = (C.c_double * 5)(*acc.tolist())
                t0c = time.perf_counter()
                bad = lib.wass_coll_init(ctx._h, rank, world, uid) != 0 or lib.wass_coll_allreduce_sum_f64(ctx._h, a5, 5) != 0
                coll_ms = (time.perf_counter() - t0c) * 1e3
                bad_t = torch.tensor([1 if bad else 0], dtype=torch.int32, device=coll_dev)
                dist.all_reduce(bad_t, op=dist.ReduceOp.MAX)
                if int(bad_t.item()):
                    coll_err = ("rank %d: " % rank + lib.wass_last_error(ctx._h).decode()) if bad else "wass_coll failed on another rank"
            if coll_err is None:
                acc = np.array(a5[:])
                coll_info = {"path": "wass_coll_allreduce_sum_f64 (RCCL via the C ABI)", "ranks": world, "ms_incl_comm_init": round(coll_ms, 2),
                             "matches_torch_distributed": bool(np.array_equal(acc, acc_torch))}
            else:
                print(f"bench.py: rank {rank}: wass_coll all-reduce FAILED: {coll_err}", file=sys.stderr, flush=True)
                acc = acc_torch
                coll_info = {"path": "torch.distributed all-reduce (the library's own RCCL path FAILED)", "ranks": world, "error": coll_err}
        else:
            acc = acc_torch
            coll_info = {"path": "torch.distributed gloo (shared-GPU functional test: RCCL refuses two ranks on one device)", "ranks": world}
    mean_plane, n_planes = wass_amd.planes_mean_finish(acc)
    if dist is not None:
        mine = torch.tensor([args.steps / elapsed], dtype=torch.float64, device=coll_dev)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        rank_rates = [float(t.item()) for t in allr]
        el = torch.tensor([elapsed], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        elapsed = float(el.item())
    overflow = max(overflows) if overflows else ctx.sgm_timings().cost_overflow

    if rank == 0:
        pairs = world * args.steps
        pairs_s = pairs / elapsed
        cells = w * h * D
        alg_bytes = cells * (2 * args.ndirs + 4)             # (2R+4) B per cell
        t_agg = float(np.mean(agg_ms)) * 1e-3
        achieved = alg_bytes / t_agg / 1e9
        traffic = measured_traffic(args.config, args.ndirs)
        # Path 2 runs inside the cost stage's vertical-sum kernel (k_vsum_col).  Two cross-checks:
        #  strict: the SAME algorithmic bytes over the aggregation time PLUS what the column paths add to that kernel -- measured
        #          in this run as the difference to the plain vertical sum on the same horizontal sums (wass_sgm_probe_vsum);
        #  with_fused_vertical_sum: that whole kernel charged to the family, its own algorithmic bytes included.
        t_vs = float(np.mean(vsum_ms)) * 1e-3
        path2_marginal = max(0.0, vsum_probe[1] - vsum_probe[0]) * 1e-3
        achieved_strict = alg_bytes / (t_agg + path2_marginal) / 1e9
        alg_fused = alg_bytes + cells * (4 + (2 if args.ndirs == 5 else 0))
        achieved_fused = alg_fused / (t_agg + t_vs) / 1e9
        t_cost = float(np.mean(cost_ms)) * 1e-3
        cost_tops = cells * COST_OPS_PER_CELL / t_cost / 1e12
        line = {
            "metric": "stereo_pairs_per_sec", "value": round(pairs_s, 4), "unit": "pairs/s",
            "mdisp_per_sec": round(pairs_s * cells / 1e6, 1),
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ranks": {"world_size_from_process_group": world, "backend": (dist.get_backend() if dist is not None else None),
                      "pairs_per_sec_per_rank": [round(x, 3) for x in rank_rates],
                      "shared_gpu": bool(args.allow_shared_gpu and world > ndev)},
            "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u16", "data": "synthetic",
            "config": {"workload": f"config {args.config}: {w}x{h} rectified pair, D={D}, {args.ndirs}-path SGBM "
                                   + ("+ disparity clean-up + triangulation + z-gap/CC + RANSAC plane + refine + xyzC encode"
                                      if args.stage == "full" else "(a1-a6 only)") + ", frame-parallel over ranks",
                       "width": w, "height": h, "num_disp": D, "ndirs": args.ndirs, "pairs_per_rank": args.steps,
                       "distinct_frames_per_rank": nf, "stage": args.stage, "tail_overlap": tail_overlap,
                       "inputs": "resident in HBM when the timed region starts" if resident_main else
                                 "pinned host memory, uploaded inside the timed region, one frame ahead (2 images per step)"},
            "value_is": "resident_inputs" if resident_main else "pcie_inclusive",
            "roofline": {"bound": "hbm", "kernel": "path aggregation family (k_rowsweep + k_ckpt + k_pairx + k_pair [+ k_sweep]), all launches "
                                                   "of one frame, side stream included",
                         # launch durations inside pipelined frames (hipEvents around every launch, 6 frames outside the timed passes);
                         # k_rowsweep and the second k_ckpt run on the side stream beside the main stream's kernels, so the sum
                         # exceeds `ms`: the bracket is main-stream time
                         "kernel_ms": kernel_ms,
                         "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4),
                         "traffic": int(traffic[0]) if traffic else None,
                         "traffic_source": traffic[1] if traffic else None,
                         "algorithmic_bytes": alg_bytes, "ms": round(t_agg * 1e3, 3),
                         # measured bytes over the measured time: how hard the memory system is driven (the guide's achievable
                         # HBM rate is ~6.3 TB/s); the gap between this and `achieved` is traffic beyond the algorithmic bytes
                         "effective_tbps": round(traffic[0] / t_agg / 1e12, 3) if traffic else None,
                         "strict": {"achieved": round(achieved_strict, 1), "frac": round(achieved_strict / HBM_PEAK_GBS, 4),
                                    "ms": round((t_agg + path2_marginal) * 1e3, 3),
                                    "vertical_sum_ms": {"plain": round(vsum_probe[0], 3), "with_column_paths": round(vsum_probe[1], 3)},
                                    "note": "same bytes; time = aggregation launches + what the column paths add to the cost stage's vertical "
                                            "sum (measured in this run: production kernel minus plain sum on the same data)"},
                         "with_fused_vertical_sum": {"achieved": round(achieved_fused, 1), "frac": round(achieved_fused / HBM_PEAK_GBS, 4),
                                                     "algorithmic_bytes": alg_fused, "ms": round((t_agg + t_vs) * 1e3, 3)}},
            "roofline_cost_volume": {"bound": "valu", "kernel": "k_prefilter + k_hsum_q + k_vsum_col", "achieved": round(cost_tops, 2),
                                     "peak": round(VALU_PK16_PEAK_TOPS, 1), "unit": "Tops/s (u16)", "frac": round(cost_tops / VALU_PK16_PEAK_TOPS, 4),
                                     "ops_per_cell": COST_OPS_PER_CELL, "ms": round(t_cost * 1e3, 3),
                                     "peak_source": "measured issue rate, scripts/micro/valu2.hip: 4.5 cycles per wave instruction per SIMD",
                                     # the stage AS BUILT is two kernels with the horizontal sums handed over through HBM (DESIGN.md 5,
                                     # "The hsum round trip"): 2 B/cell written + 2 read for hsum, 2 written for C -- against the same 8 TB/s
                                     "as_built": {"bound": "hbm", "bytes": cells * 6, "achieved": round(cells * 6 / t_cost / 1e9, 1), "peak": HBM_PEAK_GBS,
                                                  "unit": "GB/s", "frac": round(cells * 6 / t_cost / 1e9 / HBM_PEAK_GBS, 4),
                                                  "traffic": int(traffic[2]) if traffic and traffic[2] else None}},
            "stage_ms": {"prefilter": round(float(np.mean(main_pass["tm"]["pre"])), 3), "cost_volume": round(float(np.mean(cost_ms)), 3), "vertical_sum_and_path2": round(t_vs * 1e3, 3), "aggregate": round(t_agg * 1e3, 3),
                         "select": round(float(np.mean(sel_ms)), 3), "lr_check_median_crop": round(float(np.mean(main_pass["tm"]["med"])), 3), "sgm_total": round(float(np.mean(sgm_ms)), 3)},
            "mean_plane": [None if x != x else round(float(x), 9) for x in mean_plane], "planes_averaged": n_planes,
            "plane_allreduce": coll_info,
            "points_per_frame": int(np.mean(npts_hist)) if npts_hist else None,
            "xyzc_bytes_per_frame": int(np.mean(nbytes_hist)) if nbytes_hist else None,
            "cost_overflow": int(overflow),
            "repeat_check": repeat_check(planes, npts_hist, args.warmup, nf),
        }
        def pass_record(ps, steps, resident):
            return {"pairs_per_sec": round(world * steps / ps["elapsed"], 4), "ms_per_step": round(ps["elapsed"] / steps * 1e3, 3), "steps": steps,
                    "inputs": ("resident in HBM when the clock starts" if resident else
                               "both pictures of every frame uploaded from pinned host memory inside the step, one frame ahead (wass_upload_async)"),
                    "aggregate_ms": round(float(np.mean(ps["tm"]["agg"])), 3)}
        # both definitions under stable names, whichever one is `value` (the main pass's elapsed is the max over ranks)
        line["resident_inputs" if resident_main else "pcie_inclusive"] = pass_record({**main_pass, "elapsed": elapsed}, args.steps, resident_main)
        if other_pass is not None:
            line["pcie_inclusive" if resident_main else "resident_inputs"] = pass_record(other_pass, args.steps, not resident_main)
        if pass5 is not None:
            s5 = min(args.steps, 64)
            alg5 = cells * (2 * 5 + 4)
            t5 = float(np.mean(pass5["tm"]["agg"])) * 1e-3
            tr5 = measured_traffic(args.config, 5)
            p2m5 = max(0.0, vsum_probe5[1] - vsum_probe5[0]) * 1e-3
            line["mode_5path"] = {
                "workload": f"config {args.config}: {w}x{h}, D={D}, 5-path MODE_SGBM (what wass_stereo.cpp:775-777 runs), whole chain a1-a20, "
                            f"pictures uploaded inside the step, {s5} frames",
                "pairs_per_sec": round(s5 / pass5["elapsed"], 4), "ms_per_step": round(pass5["elapsed"] / s5 * 1e3, 3), "steps": s5,
                "aggregate_ms": round(t5 * 1e3, 3),
                "stage_ms": {"cost_volume": round(float(np.mean(pass5["tm"]["cost"])), 3), "aggregate": round(t5 * 1e3, 3),
                             "sgm_total": round(float(np.mean(pass5["tm"]["sgm"])), 3)},
                "roofline": {"bound": "hbm", "kernel": "k_sweep (path 1, writes S) + k_rowsweep + k_pairx<ONE> (path 2 + rows 0/4, S +=) + k_sweep (path 3, "
                                                       "selection); path 2's forward sweep rides in k_vsum_col (checkpoints only)",
                             "achieved": round(alg5 / t5 / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": round(alg5 / t5 / 1e9 / HBM_PEAK_GBS, 4), "algorithmic_bytes": alg5, "ms": round(t5 * 1e3, 3),
                             "traffic": int(tr5[0]) if tr5 else None, "traffic_source": tr5[1] if tr5 else None,
                             "effective_tbps": round(tr5[0] / t5 / 1e12, 3) if tr5 else None,
                             "kernel_ms": kernel_ms5,
                             "strict": {"achieved": round(alg5 / (t5 + p2m5) / 1e9, 1), "frac": round(alg5 / (t5 + p2m5) / 1e9 / HBM_PEAK_GBS, 4),
                                        "ms": round((t5 + p2m5) * 1e3, 3),
                                        "vertical_sum_ms": {"plain": round(vsum_probe5[0], 3), "with_column_paths": round(vsum_probe5[1], 3)},
                                        "note": "same bytes; time = aggregation launches + what the column path adds to the cost stage's vertical "
                                                "sum (production kernel minus plain sum on the same data, probed right behind this pass)"}},
                "repeat_check": repeat_check(pass5["planes"], pass5["npts"], min(args.warmup, 5), nf),
                "cost_overflow": int(max(pass5["overflows"])) if pass5["overflows"] else 0}
        if world == 1 and args.config == "B" and args.stage == "full" and not args.no_config_e:
            ctx.close()
            ctx = None
            line["config_E"] = config_e_record(dev_index, args.ndirs)
            # the headline configuration's aggregation without the frame pipelining: in the timed region the previous frame's
            # tail (clean-up, triangulation, plane fit, encoder: 1 ms of small kernels) runs underneath these kernels and costs
            # them 0.1-0.3 ms; this is the kernel family on its own, same brackets
            line["roofline"]["sgm_stage_alone"] = {k: v for k, v in config_e_record(dev_index, args.ndirs, steps=8, config="B").items()
                                                   if k in ("workload", "aggregate_ms", "cost_volume_ms", "sgm_total_ms", "roofline")}
        if world == 1 and args.config == "B" and args.stage == "full" and not args.no_cxx_driver:
            if ctx is not None:
                ctx.close()
                ctx = None
            try:
                line["cxx_driver"] = cxx_driver_record(args.ndirs)
                # the same driver on inputs that cost the host less: workdirs written by this product's own wass_prepare (stored PNG
                # blocks), and no workdirs at all (--raw: undistortion inside the frame chain) -- host_cpu_ms_per_frame is what sets
                # how many cores a GPU needs
                keep = ("pairs_per_sec", "frames", "host_cpu_ms_per_frame", "host_cores_busy", "inputs", "error")
                for name, kw in (("product_prepared_inputs", {"png_level": 0}), ("raw_inputs", {"raw": True})):
                    sub = cxx_driver_record(args.ndirs, replicate=12, **kw)
                    line["cxx_driver"][name] = {k: sub[k] for k in keep if k in sub}
            except Exception as e:                          # the headline must not depend on a scratch directory
                line.setdefault("cxx_driver", {})["error"] = f"{type(e).__name__}: {e}"
            try:
                line["wasscli_unchanged"] = wasscli_unchanged_record(args.ndirs)
            except Exception as e:
                line["wasscli_unchanged"] = {"error": f"{type(e).__name__}: {e}"}
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args.config)
    # N > 1: the PRODUCT's scaling point beside the harness's -- the shipped sequence driver with one worker per GPU, run by rank 0
    # once every rank has let go of its GPU (contexts closed, caches emptied; the ranks wait at the barrier below meanwhile)
    multi_cxx = world > 1 and args.config == "B" and args.stage == "full" and not args.no_cxx_driver
    if multi_cxx:
        ctx.close()
        ctx = None
        del pipe, dres, dring
        torch.cuda.empty_cache()
        dist.barrier()
        if rank == 0:
            shared = bool(args.allow_shared_gpu and world > ndev)
            try:
                line["cxx_driver"] = cxx_driver_record(args.ndirs, gpus=(ndev if shared else world),
                                                       procs_per_gpu=((world + ndev - 1) // ndev if shared else 1))
            except Exception as e:
                line["cxx_driver"] = {"error": f"{type(e).__name__}: {e}"}
        # the other ranks wait on the HOST (a key of the rendezvous store): an RCCL barrier would spin on their GPUs underneath
        # the driver's workers
        try:
            store = dist.distributed_c10d._get_default_store()
            if rank == 0:
                store.set("wass_cxx_driver_done", "1")
            else:
                store.wait(["wass_cxx_driver_done"])
        except Exception:
            pass
    if rank == 0:
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if ctx is not None:
        ctx.close()


if __name__ == "__main__":
    main()
