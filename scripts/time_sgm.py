"""SGM stage timing at an arbitrary size: python scripts/time_sgm.py W H D [ndirs] [frames]  (resident inputs, hipEvent stage times)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import wass_amd
from wass_amd import synth, default_sgm_params
w, h, D = (int(v) for v in sys.argv[1:4])
nd = int(sys.argv[4]) if len(sys.argv) > 4 else 8
nf = int(sys.argv[5]) if len(sys.argv) > 5 else 8
dev = torch.device("cuda", 0)
r, l = synth.make_pair_torch(w, h, D, frame_idx=1, device=dev)
p = default_sgm_params(D, ndirs=nd)
out = torch.empty((h, w), dtype=torch.int16, device=dev)
with wass_amd.Context(0) as ctx:
    agg, cost, tot = [], [], []
    for i in range(nf + 2):
        ctx.sgm_disparity_dev(r, l, p, out)
        ctx.synchronize()
        t = ctx.sgm_timings()
        if i >= 2:
            agg.append(t.aggregate_ms); cost.append(t.cost_ms); tot.append(t.total_ms)
    print(f"{w}x{h} D={D} {nd}-path: cost {np.mean(cost):.2f} ms, aggregate {np.mean(agg):.2f} ms, total {np.mean(tot):.2f} ms, overflow {ctx.sgm_timings().cost_overflow}")
