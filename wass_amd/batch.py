"""Frame-parallel driver: the MI355X replacement for wasscli's process fan-out.

The reference runs one `wass_stereo` process per frame, N at a time (cli/wasscli/wasscli.py:305-364), and joins the
per-frame `plane.txt` files into `output/planes.txt` (:341-343); the gridding stage later takes np.nanmean of that
file as the sequence's mean sea plane (gridding/wassgridsurface/wassgridsurface.py:672-678).

Here frames are sharded over ranks (one process per GPU, frame i -> rank i mod world); there is no data-path
collective.  The only exchange is Coll-1: an all-reduce of [sum a, sum b, sum c, sum d, n_valid] (5 doubles) and a
gather of the per-frame planes so that rank 0 can write planes.txt in frame order.  Backend "nccl" is RCCL on ROCm;
"gloo" is used by the CPU tests.
"""
from __future__ import annotations

from typing import Callable, Sequence

import numpy as np

from . import stereo


def shard(n_frames: int, rank: int, world: int) -> list[int]:
    """Frame indices owned by `rank` (round-robin, like wasscli's work queue)."""
    return list(range(rank, n_frames, world))


def planes_text(planes: np.ndarray) -> str:
    """planes.txt: one frame per line, the four plane.txt values joined by single spaces (wasscli.py:341-343)."""
    lines = []
    for p in np.asarray(planes, float).reshape(-1, 4):
        lines.append("nan nan nan nan" if np.isnan(p).any() else " ".join(repr(float(v)) for v in p))
    return "\n".join(lines) + ("\n" if lines else "")


def run_sequence(n_frames: int, process_frame: Callable[[int], Sequence[float]], dist=None, device=None):
    """Process this rank's share of frames and reduce the planes.

    process_frame(i) -> plane (4 floats, NaNs when the plane fit failed; wass_stereo.cpp:2101-2107).
    Returns (mean_plane[4], n_valid, all_planes[n_frames][4] on rank 0 else None).
    """
    import torch
    rank = dist.get_rank() if dist is not None else 0
    world = dist.get_world_size() if dist is not None else 1
    mine = shard(n_frames, rank, world)
    local = np.full((len(mine), 4), np.nan)
    for k, i in enumerate(mine):
        local[k] = np.asarray(process_frame(i), float)
    acc = stereo.planes_mean_accumulate(local) if len(mine) else np.zeros(5)
    planes_all = None
    if dist is not None:
        t = torch.tensor(acc, dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)                       # Coll-1
        acc = t.cpu().numpy()
        # gather per-frame planes (padded to the largest shard) for planes.txt
        per = (n_frames + world - 1) // world
        buf = torch.full((per, 4), float("nan"), dtype=torch.float64, device=device)
        if len(mine):
            buf[:len(mine)] = torch.from_numpy(local).to(buf.device)
        out = [torch.empty_like(buf) for _ in range(world)]
        dist.all_gather(out, buf)
        if rank == 0:
            planes_all = np.full((n_frames, 4), np.nan)
            for r in range(world):
                idx = shard(n_frames, r, world)
                planes_all[idx] = out[r].cpu().numpy()[:len(idx)]
    else:
        planes_all = local
    mean, n_valid = stereo.planes_mean_finish(acc)
    return mean, n_valid, planes_all
