#!/usr/bin/env python3
"""bench.py -- throughput of the wass_stereo dense-stereo hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--ndirs 5|8] [--config B|A|E]

One "step" = one synthetic rectified stereo pair pushed through the whole GPU
hot path (inputs already resident in HBM).  With N > 1 (launched by
torch.distributed.run, one rank per GPU) every rank processes its own frames
-- stereo frames are independent, there is no data-path collective -- and the
reported value is the whole-job rate (weak scaling).

Prints ONE JSON line (rank 0) with BASELINE.json's metric plus
  roofline     -- aggregation kernel family vs the 8 TB/s HBM roofline
  cpu_baseline -- the CPU oracle timed on this box's host cores (rank 0, N=1)
"""
from __future__ import annotations

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


def cpu_baseline(w: int, h_full: int, D: int, ndirs: int, budget_cells: float = 1.0e9):
    """Time the CPU oracle (scalar C restatement, 1 thread) on a bounded band of the same workload: same width and
    disparity range, as many rows as ~15 s of CPU work allow (about 1e9 pixel-disparity cells)."""
    from oracle import oracle as O
    from wass_amd import synth
    h = int(min(h_full, max(64, budget_cells // (w * D))))
    right, left = synth.make_pair(w, h, D, frame_idx=1000)
    p = O.wass_params(D, mode=ndirs)
    t0 = time.perf_counter()
    O.dense_disparity16(right, left, p)
    dt = time.perf_counter() - t0
    mdisp = w * h * D / 1e6 / dt
    return {"value": round(mdisp, 2), "unit": "Mdisp/s", "cores": 1, "kind": "port",
            "pairs_per_sec_equivalent": round(mdisp * 1e6 / (w * h_full * D), 4),
            "sample": f"{w}x{h} band of the {w}x{h_full} workload, D={D}, {ndirs}-path SGBM stage (a1-a6), scalar C oracle, "
                      f"1 thread, {dt:.1f}s"}


def measured_traffic(config: str, ndirs: int):
    """Per-frame HBM bytes of the aggregation kernels from the committed rocprofv3 PMC summary of this command
    (profiles/*traffic_<config>_<ndirs>path.json, produced by scripts/profile.sh + scripts/traffic_json.py)."""
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", f"*traffic_{config}_{ndirs}path.json"))):
        try:
            j = json.load(open(f))
            if j.get("config") == config and j.get("ndirs") == ndirs:
                best = (j["aggregation_hbm_bytes_per_frame"], os.path.relpath(f, ROOT))
        except Exception:
            pass
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--ndirs", type=int, default=8, choices=(5, 8))
    ap.add_argument("--config", default="B", choices=sorted(CONFIGS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-tail-overlap", action="store_true",
                    help="run the post-SGM stages on the SGM stream instead of the context's tail stream")
    ap.add_argument("--inflight", type=int, default=1,
                    help="frames in flight per GPU (each on its own context/stream/scratch, one host thread each)")
    ap.add_argument("--stage", default="full", choices=("full", "sgm"),
                    help="full = a1-a20 (SGBM, clean-up, triangulation, plane fit, xyzC); sgm = a1-a6 only")
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
    if local_rank >= ndev:
        if not args.allow_shared_gpu:
            sys.exit(f"bench.py: rank {rank} needs GPU {local_rank} but only {ndev} device(s) are visible "
                     f"(--allow-shared-gpu runs the ranks on shared devices, for functional tests only)")
    dev_index = local_rank % ndev
    torch.cuda.set_device(dev_index)
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
    nslot = max(1, args.inflight)
    ctxs = [wass_amd.Context(dev_index) for _ in range(nslot)]
    ctx = ctxs[0]
    tail_overlap = args.stage == "full" and not args.no_tail_overlap
    for c_ in ctxs:
        c_.set_tail_overlap(tail_overlap)

    # two different resident frames per rank, alternated, so no step can reuse a previous result
    frames = []
    for k in range(2):
        r, l = synth.make_pair(w, h, D, frame_idx=rank * 16 + k)
        frames.append((torch.from_numpy(r).to(dev), torch.from_numpy(l).to(dev)))
    geom = wass_amd.make_geom(synth.rig_geometry(w, h))
    burned = [(fr[0] <= 254).to(torch.uint8) for fr in frames]      # DISCARD_BURNED_AREAS masks (right image)
    planes, npts_hist, nbytes_hist = [], [], []
    # wass_stereo.cpp main() per frame: SGM -> clean-up -> triangulate -> z-gap / biggest component -> RANSAC -> crop ->
    # refine -> crop -> mesh_cam.xyzC (defaults of SURVEY.md Appendix C, RANDOM_SEED=12345), as wass_amd.batch.FramePipeline
    # enqueues it: no host synchronisation inside a frame, the previous frame's output is collected while this one runs
    from wass_amd.batch import FramePipeline
    pipes = [FramePipeline(c_, w, h, params, geom, tail_overlap=tail_overlap) for c_ in ctxs] if args.stage == "full" else []
    sgm_out = [torch.empty((h, w), dtype=torch.int16, device=dev) for _ in range(nslot)]

    def keep(o):
        if o is not None:
            planes.append(o.plane); npts_hist.append(o.n_points); nbytes_hist.append(len(o.xyzc))

    def step(i, slot=0):
        dr, dl = frames[i % 2]
        if args.stage == "sgm":
            ctxs[slot].sgm_disparity_dev(dr, dl, params, sgm_out[slot])
        else:
            keep(pipes[slot].submit(dr, dl, d_right_image=dr, d_right_mask=burned[i % 2]))

    def barrier():
        for p_ in pipes:
            keep(p_.flush())
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(max(args.warmup, nslot)):
        step(i, i % nslot)
    barrier()
    planes.clear(); npts_hist.clear(); nbytes_hist.clear()
    agg_ms, cost_ms, sel_ms, sgm_ms, vsum_ms = [], [], [], [], []

    def take(t):
        agg_ms.append(t.aggregate_ms); cost_ms.append(t.cost_ms); sel_ms.append(t.select_ms); sgm_ms.append(t.total_ms)
        vsum_ms.append(t.vsum_ms)

    def run_slot(slot):
        # frames slot, slot+nslot, ... : each slot is an independent context (streams + scratch HBM).  Stage timings
        # come from hipEvents recorded on the context's own stream; frame n's are read after frame n+1 has been
        # enqueued (two event sets), so the reader never drains the pipeline
        mine = list(range(slot, args.steps, nslot))
        for k, i in enumerate(mine):
            step(i, slot)
            if k > 0:
                take(ctxs[slot].sgm_timings(previous=True))
        if mine:
            take(ctxs[slot].sgm_timings())

    t0 = time.perf_counter()
    if nslot == 1:
        run_slot(0)
    else:
        import threading
        th = [threading.Thread(target=run_slot, args=(s_,)) for s_ in range(nslot)]
        for t_ in th: t_.start()
        for t_ in th: t_.join()
    barrier()
    elapsed = time.perf_counter() - t0
    # Coll-1: sequence mean plane = NaN-aware mean over every rank's frames (5 doubles all-reduced over RCCL)
    acc = wass_amd.planes_mean_accumulate(np.array(planes).reshape(-1, 4)) if planes else np.zeros(5)
    rank_rates = [args.steps / elapsed]
    if dist is not None:
        acc_t = torch.tensor(acc, dtype=torch.float64, device=coll_dev)
        dist.all_reduce(acc_t, op=dist.ReduceOp.SUM)
        acc = acc_t.cpu().numpy()
    mean_plane, n_planes = wass_amd.planes_mean_finish(acc)
    if dist is not None:
        mine = torch.tensor([args.steps / elapsed], dtype=torch.float64, device=coll_dev)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        rank_rates = [float(t.item()) for t in allr]
        el = torch.tensor([elapsed], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        elapsed = float(el.item())
    overflow = ctx.sgm_timings().cost_overflow

    if rank == 0:
        pairs = world * args.steps
        pairs_s = pairs / elapsed
        cells = w * h * D
        alg_bytes = cells * (2 * args.ndirs + 4)             # (2R+4) B per cell
        t_agg = float(np.mean(agg_ms)) * 1e-3
        achieved = alg_bytes / t_agg / 1e9
        traffic = measured_traffic(args.config, args.ndirs)
        # Path 2 runs inside the cost stage's vertical-sum kernel (k_vsum_col).  Conservative cross-check that charges
        # that whole kernel to the family: its time is added and so are its own algorithmic bytes (hsum read + C write,
        # 4 B/cell, + the S = L_2 write of the 5-path mode, 2 B/cell).
        t_vs = float(np.mean(vsum_ms)) * 1e-3
        alg_fused = alg_bytes + cells * (4 + (2 if args.ndirs == 5 else 0))
        achieved_fused = alg_fused / (t_agg + t_vs) / 1e9
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
                       "stage": args.stage, "frames_in_flight": nslot, "tail_overlap": tail_overlap},
            "roofline": {"bound": "hbm", "kernel": "path aggregation family (k_ckpt + k_pair [+ k_sweep]), all launches of one frame",
                         "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4),
                         "traffic": int(traffic[0]) if traffic else None,
                         "traffic_source": traffic[1] if traffic else None,
                         "algorithmic_bytes": alg_bytes, "ms": round(t_agg * 1e3, 3),
                         "with_fused_vertical_sum": {"achieved": round(achieved_fused, 1), "frac": round(achieved_fused / HBM_PEAK_GBS, 4),
                                                     "algorithmic_bytes": alg_fused, "ms": round((t_agg + t_vs) * 1e3, 3)}},
            "stage_ms": {"cost_volume": round(float(np.mean(cost_ms)), 3), "vertical_sum_and_path2": round(t_vs * 1e3, 3), "aggregate": round(t_agg * 1e3, 3),
                         "select": round(float(np.mean(sel_ms)), 3), "sgm_total": round(float(np.mean(sgm_ms)), 3)},
            "mean_plane": [None if x != x else round(float(x), 9) for x in mean_plane], "planes_averaged": n_planes,
            "points_per_frame": int(np.mean(npts_hist)) if npts_hist else None,
            "xyzc_bytes_per_frame": int(np.mean(nbytes_hist)) if nbytes_hist else None,
            "cost_overflow": int(overflow),
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(w, h, D, args.ndirs)
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    for c_ in ctxs:
        c_.close()


if __name__ == "__main__":
    main()
