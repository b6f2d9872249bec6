"""How fast is the existing pair schedule on a SHORT image whose S and C volumes fit the 256 MiB Infinity Cache?
(An estimate for a banded schedule: rows / cols / diag / anti over one band of rows back to back.)
  python scripts/band_probe.py            -> ns per cell of the aggregation stage for several heights at w = 2456, D = 256"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import wass_amd
from wass_amd import synth

ctx = wass_amd.Context(0)
w, D = 2456, 256
for h in (16, 32, 64, 96, 128, 192, 256, 512, 2058):
    right, left = synth.make_pair(w, max(h, 64), D, frame_idx=0)
    right, left = right[:h], left[:h]
    p = wass_amd.default_sgm_params(num_disp=D, ndirs=8)
    dl = torch.from_numpy(left).cuda(); dr = torch.from_numpy(right).cuda()
    out = torch.empty((h, w), dtype=torch.int16, device="cuda")
    agg, vs, tot = [], [], []
    for it in range(12):
        ctx.sgm_disparity_dev(dr, dl, p, out)
        torch.cuda.synchronize()
        t = ctx.sgm_timings()
        if it >= 4:
            agg.append(t.aggregate_ms); vs.append(t.vsum_ms); tot.append(t.total_ms)
    cells = (w - 0) * h * D
    a = float(np.mean(agg))
    print(f"h={h:5d}  agg {a:8.3f} ms  {a * 1e6 / cells:7.4f} ns/cell  (full-frame equivalent {a * 1e6 / cells * 2456 * 2058 * 256 / 1e6:6.2f} ms)  "
          f"vsum {np.mean(vs):.3f}  total {np.mean(tot):.3f}  S band {cells * 2 / 2**20:.0f} MiB", flush=True)
