"""Aggregation-time variance across contexts (fresh HBM allocations) within one process."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
import wass_amd
from wass_amd import synth

w, h, D = 2456, 2058, 256
p = wass_amd.default_sgm_params(D, ndirs=8)
r, l = synth.make_pair(w, h, D, frame_idx=0)
dr, dl = torch.from_numpy(r).cuda(), torch.from_numpy(l).cuda()
out = torch.empty((h, w), dtype=torch.int16, device="cuda")
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 4):
    ctx = wass_amd.Context(0)
    ts = []
    for i in range(8):
        ctx.sgm_disparity_dev(dr, dl, p, out)
        t = ctx.sgm_timings()
        if i >= 2:
            ts.append((t.cost_ms, t.aggregate_ms, t.total_ms))
    a = np.array(ts)
    print("ctx %d: cost %.3f agg %.3f (min %.3f max %.3f) total %.3f" % (rep, a[:, 0].mean(), a[:, 1].mean(), a[:, 1].min(), a[:, 1].max(), a[:, 2].mean()), flush=True)
    ctx.close()
    junk = torch.empty(int(1e9 * (rep + 1)) // 3, dtype=torch.uint8, device="cuda")   # perturb the allocator
