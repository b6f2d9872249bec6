#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (kernel stats + FETCH_SIZE / WRITE_SIZE passes) into a small text summary."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]


def find(sub, pat):
    r = glob.glob(os.path.join(out, sub, "**", pat), recursive=True)
    return r[0] if r else None


def short(name):
    name = name.split("(")[0]
    return name.replace("void wass::", "").replace("wass::", "")[:70]


st = find("stats", "*kernel_stats.csv")
if st:
    print("== rocprofv3 --kernel-trace --stats (per kernel) ==")
    print(f"{'kernel':70s} {'calls':>6s} {'total_ms':>10s} {'avg_us':>10s} {'pct':>6s}")
    for r in csv.DictReader(open(st)):
        print(f"{short(r['Name']):70s} {r['Calls']:>6s} {float(r['TotalDurationNs'])/1e6:10.3f} "
              f"{float(r['AverageNs'])/1e3:10.1f} {float(r['Percentage']):6.2f}")
for sub, ctr in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
    f = find(sub, "*counter_collection.csv")
    if not f:
        continue
    acc = defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        if r.get("Counter_Name") != ctr:
            continue
        a = acc[short(r["Kernel_Name"])]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
    print(f"\n== --pmc {ctr} (KiB units as reported; per launch average) ==")
    for k, (n, v) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
        print(f"{k:70s} launches {n:5d}  avg {v/n/1024:12.2f} MiB/launch   total {v/1024/1024:10.3f} GiB")
