#!/usr/bin/env python3
"""Timeline of the aggregation kernels of the last profiled frame: scripts/frame_timeline.py gpurun_out/prof_<tag>"""
import csv, glob, os, sys
f = glob.glob(os.path.join(sys.argv[1], "stats", "**", "*kernel_trace.csv"), recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if any(k in r["Kernel_Name"] for k in ("sweep", "k_tile", "vsum", "k_pair", "k_ckpt", "hsum"))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
last = max(i for i, r in enumerate(rows) if "hsum" in r["Kernel_Name"])
t0 = int(rows[last]["Start_Timestamp"])
for r in rows[last:]:
    print(f"{r['Kernel_Name'].replace('void wass::','')[:34]:34s} {(int(r['Start_Timestamp'])-t0)/1e6:8.3f} -> {(int(r['End_Timestamp'])-t0)/1e6:8.3f} ms  vgpr {r.get('VGPR_Count')} grid {r.get('Grid_Size_X') or r.get('Grid_Size')}")
