#!/bin/bash
# Round-3 profile set on one box: stats + FETCH/WRITE passes for config B (8- and 5-path) and config E, SQ counters for B.
#   scripts/r03_profiles.sh <tag>
TAG=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
scripts/profile.sh ${TAG}_B8 > /dev/null 2>&1
scripts/traffic_json.py gpurun_out/prof_${TAG}_B8 B 8 > gpurun_out/${TAG}_traffic_B_8path.json
scripts/profile.sh ${TAG}_B5 --ndirs 5 > /dev/null 2>&1
scripts/traffic_json.py gpurun_out/prof_${TAG}_B5 B 5 > gpurun_out/${TAG}_traffic_B_5path.json
scripts/profile.sh ${TAG}_E8 --config E > /dev/null 2>&1
scripts/traffic_json.py gpurun_out/prof_${TAG}_E8 E 8 > gpurun_out/${TAG}_traffic_E_8path.json
scripts/pmc_sq.sh ${TAG}_sq --no-tail-overlap > gpurun_out/${TAG}_sq_counters.txt 2>&1
for c in B8 B5 E8; do cp gpurun_out/prof_${TAG}_$c/summary.txt gpurun_out/${TAG}_${c}_summary.txt; cp $(find gpurun_out/prof_${TAG}_$c/stats -name "*kernel_stats.csv" | head -1) gpurun_out/${TAG}_${c}_kernel_stats.csv; done
python bench.py > gpurun_out/${TAG}_bench_B_default.log 2> gpurun_out/${TAG}_bench_B_default.err
python bench.py --config E --steps 20 --warmup 4 --no-cpu-baseline > gpurun_out/${TAG}_bench_E.log 2>/dev/null
python bench.py --ndirs 5 --no-cpu-baseline --no-config-e > gpurun_out/${TAG}_bench_B_5path.log 2>/dev/null
grep -h -o '"aggregation_hbm_bytes_per_frame": [0-9.e+]*' gpurun_out/${TAG}_traffic_*.json
tail -c 400 gpurun_out/${TAG}_bench_B_default.log
