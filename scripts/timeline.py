"""Per-frame kernel timeline from a rocprofv3 kernel trace: python scripts/timeline.py <run_kernel_trace.csv> [frame_from_end]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "k_prefilter" in r["Kernel_Name"]][::2]
k = int(sys.argv[2]) if len(sys.argv) > 2 else 2
s, e = idx[-k], idx[-k + 1] if k > 1 else len(rows)
t0 = int(rows[s]["Start_Timestamp"])
for r in rows[s:e]:
    st = int(r["Start_Timestamp"]) - t0
    en = int(r["End_Timestamp"]) - t0
    name = r["Kernel_Name"].replace("void wass::", "").replace("wass::", "").split("(")[0][:32]
    if en - st > 20000 or "--all" in sys.argv:
        print(f"{st / 1e3:9.1f} {en / 1e3:9.1f} {(en - st) / 1e3:8.1f}  q={r['Queue_Id']} {name}")
print("frame period %.3f ms" % ((int(rows[e]["Start_Timestamp"]) - t0) / 1e6 if e < len(rows) else 0))
