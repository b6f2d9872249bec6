#!/bin/bash
# aggregation time for different checkpoint spacings K (build-time), on the GPU box: scripts/ktune.sh <macro> <config> k1 k2 ...
cd ${GRAFT_REPO_ROOT:-.}
M=$1; CFG=$2; shift 2
for k in "$@"; do
  WASS_EXTRA_FLAGS="-D$M=$k" python -m wass_amd.build --force > /dev/null 2>&1
  for rep in 1 2; do
  echo "$M=$k: $(python bench.py --steps 6 --warmup 2 --no-cpu-baseline --config $CFG --stage sgm | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["stage_ms"], d["roofline"]["frac"])')"
  done
done
python -m wass_amd.build --force > /dev/null 2>&1
