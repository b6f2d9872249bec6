#!/bin/bash
# kernel-trace stats only (fast): scripts/quickstats.sh <tag> [bench args]
TAG=${1:-q}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG; mkdir -p "$OUT"
export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o run -- python "$ROOT/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-config-e $* > "$OUT/stats.log" 2>&1
python "$ROOT/scripts/summarize_prof.py" "$OUT" | head -16
tail -1 "$OUT/stats.log" | cut -c1-400
