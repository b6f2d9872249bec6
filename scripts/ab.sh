#!/bin/bash
# A/B on one box: scripts/ab.sh <out-tag> [bench args]  -- alternates the round-2 library and the current one
TAG=${1:-ab}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG; mkdir -p "$OUT"
cd "$ROOT"
for i in 1 2; do
  WASS_GPU_LIB=$ROOT/scripts/libwassgpu_r02.so python bench.py --steps 40 --warmup 8 --no-cpu-baseline "$@" > "$OUT/base_$i.json" 2> "$OUT/base_$i.err"
  python bench.py --steps 40 --warmup 8 --no-cpu-baseline "$@" > "$OUT/new_$i.json" 2> "$OUT/new_$i.err"
done
python - "$OUT" <<'PY'
import json, sys, glob, os
for f in sorted(glob.glob(os.path.join(sys.argv[1], "*.json"))):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), "pairs/s", round(j["value"], 2), "ms", round(j["ms_per_step"], 3), "roofline", j.get("roofline", {}).get("frac"), "agg_ms", j.get("stage_ms", j.get("timings", {})))
    except Exception as e:
        print(os.path.basename(f), "ERR", e, open(f.replace(".json", ".err")).read()[-400:])
PY
