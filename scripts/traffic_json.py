#!/usr/bin/env python3
"""HBM traffic of the aggregation kernel family from the two --pmc passes of scripts/profile.sh.

    scripts/traffic_json.py gpurun_out/prof_<tag> <config> <ndirs> > profiles/<name>.json

FETCH_SIZE / WRITE_SIZE are reported in KiB.  Per MI355X_MICROARCH.md (HBM section) FETCH_SIZE counts a wide
coalesced read stream at exactly half its bytes on gfx950 -- calibrated on our own pattern (profiles/README.md) --
so reads are doubled; WRITE_SIZE is used as reported.
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

out, config, ndirs = sys.argv[1], sys.argv[2], int(sys.argv[3])
AGG = ("k_ckpt", "k_pair", "k_sweep", "k_rowsweep")          # k_pair also matches k_pairx


def per_kernel(sub, ctr):
    f = glob.glob(os.path.join(out, sub, "**", "*counter_collection.csv"), recursive=True)[0]
    acc = defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == ctr:
            k = r["Kernel_Name"].split("(")[0].replace("void wass::", "").replace("wass::", "")
            acc[k][0] += 1; acc[k][1] += float(r["Counter_Value"])
    return acc


fetch, write = per_kernel("pmc_fetch", "FETCH_SIZE"), per_kernel("pmc_write", "WRITE_SIZE")
frames = None
rows = {}
tot_r = tot_w = 0.0
for k in sorted(set(fetch) | set(write)):
    if not k.startswith(AGG):
        continue
    n = fetch[k][0]
    rows[k] = {"launches": n, "read_bytes_per_launch": 2 * fetch[k][1] / n * 1024, "write_bytes_per_launch": write[k][1] / max(write[k][0], 1) * 1024}
# frames = launches of the horizontal cost sum (one per SGM call)
hs = [k for k in fetch if k.startswith("k_hsum_q")]
frames = fetch[hs[0]][0] if hs else 1
for k, v in rows.items():
    tot_r += v["read_bytes_per_launch"] * v["launches"] / frames
    tot_w += v["write_bytes_per_launch"] * v["launches"] / frames
# the cost stage (not part of the aggregation figure): the production launches only -- k_vsum_col runs 10 times in 7 frames
# because wass_sgm_probe_vsum re-runs it three times after the timed region, with the same bytes per launch
cost = {}
cr = cw = 0.0
for k in sorted(set(fetch) | set(write)):
    if k.startswith(("k_prefilter", "k_hsum_q", "k_vsum_col")):
        n = fetch[k][0]
        cost[k] = {"launches": n, "read_bytes_per_launch": 2 * fetch[k][1] / n * 1024, "write_bytes_per_launch": write[k][1] / max(write[k][0], 1) * 1024}
        cr += cost[k]["read_bytes_per_launch"]
        cw += cost[k]["write_bytes_per_launch"]
print(json.dumps({"config": config, "ndirs": ndirs, "frames_profiled": frames, "aggregation_read_bytes_per_frame": tot_r,
                  "cost_stage_read_bytes_per_frame": cr, "cost_stage_write_bytes_per_frame": cw, "cost_stage_kernels": cost,
                  "aggregation_write_bytes_per_frame": tot_w, "aggregation_hbm_bytes_per_frame": tot_r + tot_w,
                  "fetch_size_correction": 2.0, "kernels": rows,
                  "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes, scripts/profile.sh"}, indent=1))
