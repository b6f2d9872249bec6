"""Two (or N) contexts on ONE GPU, one host thread each, every one running the full frame pipeline back to back on frames
resident in HBM: does the SGM stage of one frame fill the holes of another's?  python scripts/two_ctx.py [nctx] [frames]"""
import sys, os, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import wass_amd
from wass_amd import synth, default_sgm_params
from wass_amd.batch import FramePipeline

nctx = int(sys.argv[1]) if len(sys.argv) > 1 else 2
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 40
stage = sys.argv[3] if len(sys.argv) > 3 else "full"
w, h, D = 2456, 2058, 256
params = default_sgm_params(D, ndirs=8)
geom = wass_amd.make_geom(synth.rig_geometry(w, h))
dev = torch.device("cuda", 0)
pairs = []
for i in range(2):
    r, l = synth.make_pair(w, h, D, frame_idx=i)
    pairs.append((torch.from_numpy(r).to(dev), torch.from_numpy(l).to(dev)))

def worker(idx, n, out):
    ctx = wass_amd.Context(0)
    if stage == "full":
        pipe = FramePipeline(ctx, w, h, params, geom)
        step = lambda i: pipe.submit(*pairs[i & 1])
        fin = pipe.flush
    else:
        outs = [torch.empty((h, w), dtype=torch.int16, device=dev) for _ in range(2)]
        step = lambda i: ctx.sgm_disparity_dev(pairs[i & 1][0], pairs[i & 1][1], params, outs[i & 1])
        fin = lambda: None
    for i in range(4): step(i)
    fin(); ctx.synchronize()
    out["ready"].wait()
    t0 = time.perf_counter()
    for i in range(n): step(i)
    fin(); ctx.synchronize()
    out[idx] = time.perf_counter() - t0
    ctx.close()

for nc in ([1, nctx] if nctx > 1 else [1]):
    res = {"ready": threading.Event()}
    ths = [threading.Thread(target=worker, args=(k, frames, res)) for k in range(nc)]
    for t in ths: t.start()
    time.sleep(8 if nc > 1 else 5)
    t0 = time.perf_counter(); res["ready"].set()
    for t in ths: t.join()
    wall = time.perf_counter() - t0
    print(f"{nc} context(s) x {frames} frames ({stage}): wall {wall*1e3:.1f} ms -> {nc*frames/wall:.1f} pairs/s  per-thread {[round(res[k]*1e3,1) for k in range(nc)]}", flush=True)
