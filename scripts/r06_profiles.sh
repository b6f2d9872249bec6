#!/bin/bash
# Round-6 profile set on one box: stats + FETCH/WRITE passes for config B (8- and 5-path) and config E (row-fused now), the
# driver-style bench line, the config E / 5-path bench lines.
#   scripts/r06_profiles.sh <tag>
TAG=${1:-r06}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
scripts/profile.sh ${TAG}_B8 > /dev/null 2>&1
scripts/traffic_json.py gpurun_out/prof_${TAG}_B8 B 8 > gpurun_out/${TAG}_traffic_B_8path.json
scripts/profile.sh ${TAG}_E8 --config E > /dev/null 2>&1
scripts/traffic_json.py gpurun_out/prof_${TAG}_E8 E 8 > gpurun_out/${TAG}_traffic_E_8path.json
scripts/profile.sh ${TAG}_B5 --ndirs 5 > /dev/null 2>&1
scripts/traffic_json.py gpurun_out/prof_${TAG}_B5 B 5 > gpurun_out/${TAG}_traffic_B_5path.json
for c in B8 B5 E8; do cp gpurun_out/prof_${TAG}_$c/summary.txt gpurun_out/${TAG}_${c}_summary.txt; cp $(find gpurun_out/prof_${TAG}_$c/stats -name "*kernel_stats.csv" | head -1) gpurun_out/${TAG}_${c}_kernel_stats.csv; done
# the traffic files the bench lines quote must be in profiles/ before the bench runs
cp gpurun_out/${TAG}_traffic_*.json profiles/
python bench.py --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_driver_args.json 2> gpurun_out/${TAG}_bench_driver_args.err
python bench.py --config E --steps 20 --warmup 4 --no-cpu-baseline > gpurun_out/${TAG}_bench_E.log 2>/dev/null
python bench.py --ndirs 5 --no-cpu-baseline --no-config-e --no-cxx-driver > gpurun_out/${TAG}_bench_B_5path.log 2>/dev/null
grep -h -o '"aggregation_hbm_bytes_per_frame": [0-9.e+]*' gpurun_out/${TAG}_traffic_*.json
tail -c 600 gpurun_out/${TAG}_bench_driver_args.json
