#!/bin/bash
# SQ-level counters for every kernel of one bench run: scripts/pmc_sq.sh <tag> [bench args]
TAG=${1:-sq}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG; mkdir -p "$OUT"
export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SMEM --output-format csv -d "$OUT/pmc_sq" -o run -- python "$ROOT/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-config-e --no-cxx-driver --no-pcie-pass $* > "$OUT/pmc_sq.log" 2>&1
python - "$OUT" <<'PY'
import csv, glob, os, sys
from collections import defaultdict
out = sys.argv[1]
f = glob.glob(os.path.join(out, "pmc_sq", "**", "*counter_collection.csv"), recursive=True)[0]
acc = defaultdict(lambda: defaultdict(float)); n = defaultdict(int)
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0].replace("void wass::", "").replace("wass::", "")[:40]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_WAVES": n[k] += 1
cols = ["SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_INSTS_VALU", "SQ_INSTS_SMEM"]
print(f"{'kernel':40s} launches " + " ".join(f"{c[3:]:>14s}" for c in cols))
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]["SQ_WAVE_CYCLES"]):
    print(f"{k:40s} {n[k]:8d} " + " ".join(f"{v[c]/max(n[k],1):14.3e}" for c in cols))
PY
