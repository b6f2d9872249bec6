"""A/B of two builds of libwassgpu.so on ONE box, alternating: python scripts/ab_cost.py <other.so> [rounds]
Prints cost-stage / hsum / aggregation times of the SGM stage at config B (resident inputs) per build and round."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, os
sys.path.insert(0, %r)
import torch, numpy as np
import wass_amd
from wass_amd import synth, default_sgm_params
w, h, D = 2456, 2058, 256
dev = torch.device("cuda", 0)
r, l = synth.make_pair_torch(w, h, D, frame_idx=1, device=dev)
p = default_sgm_params(D, ndirs=int(os.environ.get("AB_NDIRS", "8")))
out = torch.empty((h, w), dtype=torch.int16, device=dev)
with wass_amd.Context(0) as ctx:
    pre, cost, vs, agg, tot = [], [], [], [], []
    for i in range(14):
        ctx.sgm_disparity_dev(r, l, p, out); ctx.synchronize()
        t = ctx.sgm_timings()
        if i >= 4: pre.append(t.prefilter_ms); cost.append(t.cost_ms); vs.append(t.vsum_ms); agg.append(t.aggregate_ms); tot.append(t.total_ms)
    print("pre %%.3f cost %%.3f (hsum %%.3f vsum %%.3f) agg %%.3f total %%.3f  csum %%d" %% (np.mean(pre), np.mean(cost), np.mean(cost) - np.mean(vs), np.mean(vs), np.mean(agg), np.mean(tot), int(out.to(torch.int64).sum())))
''' % ROOT
other = sys.argv[1]
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
for i in range(rounds):
    for name, lib in (("base", None), ("other", other)):
        env = dict(os.environ)
        if lib: env["WASS_GPU_LIB"] = os.path.abspath(lib)
        else: env.pop("WASS_GPU_LIB", None)
        o = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
        print(name, (o.stdout.strip().splitlines() or [o.stderr[-300:]])[-1], flush=True)
