"""Multi-process (world_size 2, gloo, CPU) test of the frame-parallel driver and the plane all-reduce (Coll-1)."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, n_frames, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    from wass_amd import batch
    dist.init_process_group("gloo", rank=rank, world_size=world)
    seen = []

    def process(i):                       # a stand-in for the per-frame GPU chain: deterministic plane per frame
        seen.append(i)
        if i % 5 == 3:
            return [float("nan")] * 4     # RANSAC failed on this frame (plane.txt = "nan nan nan nan")
        return [0.01 * i, -0.4 + 0.001 * i, 0.9, -11.0 - 0.1 * i]
    mean, n_valid, planes = batch.run_sequence(n_frames, process, dist=dist)
    q.put((rank, seen, mean.tolist(), n_valid, None if planes is None else planes.tolist()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_frames", [7, 2, 1])
def test_two_rank_sequence_matches_numpy_nanmean(n_frames):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_frames, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    expect = np.array([[np.nan] * 4 if i % 5 == 3 else [0.01 * i, -0.4 + 0.001 * i, 0.9, -11.0 - 0.1 * i] for i in range(n_frames)])
    # sharding: disjoint, complete, round-robin
    assert sorted(res[0][1] + res[1][1]) == list(range(n_frames))
    assert res[0][1] == list(range(0, n_frames, 2)) and res[1][1] == list(range(1, n_frames, 2))
    nm = np.nanmean(expect, axis=0)
    for rank, seen, mean, n_valid, planes in res:
        np.testing.assert_allclose(mean, nm, rtol=1e-14)              # every rank ends with the same mean plane
        assert n_valid == int((~np.isnan(expect[:, 0])).sum())
    np.testing.assert_array_equal(np.array(res[0][4]), expect)        # rank 0 has all planes in frame order
    assert res[1][4] is None


def test_planes_text_format():
    sys.path.insert(0, ROOT)
    from wass_amd import batch
    z = np.load(os.path.join(ROOT, "tests", "golden", "planes_txt.npz"))
    txt = batch.planes_text(z["planes"])
    lines = txt.strip().split("\n")
    assert lines[1] == "nan nan nan nan" and len(lines) == 3
    back = np.array([[float(x) for x in l.split()] for l in lines])
    np.testing.assert_array_equal(np.nan_to_num(back, nan=-1), np.nan_to_num(z["planes"], nan=-1))
    assert batch.shard(7, 1, 3) == [1, 4]
