#!/bin/bash
# scripts/ab_envn.sh VAR "v1 v2 v3 ..." [rounds] [bench args] -- alternate several values of an environment variable under bench.py on one box
VAR=$1; VALS=$2; N=${3:-2}; shift 3
for i in $(seq $N); do
  for v in $VALS; do
    env $VAR=$v python bench.py --steps 64 --warmup 8 --no-cpu-baseline --no-cxx-driver --no-pcie-pass --no-5path "$@" 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$VAR=$v', j['value'], j['ms_per_step'], j['stage_ms']['cost_volume'], j['stage_ms']['aggregate'], 'alone', j['roofline'].get('sgm_stage_alone',{}).get('aggregate_ms'))"
  done
done
