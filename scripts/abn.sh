#!/bin/bash
# scripts/abn.sh <tag> "<lib1> <lib2> ..." [bench args] -- alternate several builds of the library on one box
TAG=$1; LIBS=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG; mkdir -p "$OUT"; cd "$ROOT"
for i in 1 2; do for L in $LIBS; do
  n=$(basename $L .so)
  WASS_GPU_LIB=$ROOT/$L python bench.py --steps 40 --warmup 8 --no-cpu-baseline "$@" > "$OUT/${n}_$i.json" 2> "$OUT/${n}_$i.err"
done; done
python - "$OUT" <<'PY'
import json, sys, glob, os
for f in sorted(glob.glob(os.path.join(sys.argv[1], "*.json"))):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
        print(f"{os.path.basename(f):28s} pairs/s {j['value']:7.2f} ms {j['ms_per_step']:7.3f} frac {j.get('roofline', {}).get('frac')}", j.get("stage_ms"))
    except Exception as e:
        print(os.path.basename(f), "ERR", e, open(f.replace(".json", ".err")).read()[-300:])
PY
