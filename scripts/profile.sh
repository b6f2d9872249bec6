#!/bin/bash
# Profile bench.py on the GPU box with rocprofv3.
#   scripts/profile.sh <tag> [bench args...]
# Writes gpurun_out/prof_<tag>/{stats,pmc_fetch,pmc_write}/ ; copy the summaries you want judged into profiles/.
set -u
TAG=${1:-r01}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
ARGS="--steps 5 --warmup 2 --no-cpu-baseline --no-config-e --no-cxx-driver --no-pcie-pass --no-5path $*"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o run -- python "$ROOT/bench.py" $ARGS > "$OUT/stats.log" 2>&1
# counters in their own passes (no trace domains besides --kernel-trace)
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -o run -- python "$ROOT/bench.py" $ARGS > "$OUT/pmc_fetch.log" 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -o run -- python "$ROOT/bench.py" $ARGS > "$OUT/pmc_write.log" 2>&1
find "$OUT" -name "*.csv" | head -20
python "$ROOT/scripts/summarize_prof.py" "$OUT" > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
