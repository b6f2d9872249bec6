#!/bin/bash
# Alternate two values of an environment variable under bench.py (pipelined full chain) on ONE box:
#   scripts/ab_env.sh VAR A B [rounds] [bench args]        e.g.  scripts/ab_env.sh WASS_DIAG_FUSE 0 1 4
VAR=$1; A=$2; B=$3; N=${4:-3}; shift 4
for i in $(seq $N); do
  for v in "$A" "$B"; do
    env $VAR=$v python bench.py --steps 64 --warmup 8 --no-cpu-baseline --no-cxx-driver --no-config-e --no-pcie-pass --no-5path "$@" 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$VAR=$v', j['value'], j['ms_per_step'], j['stage_ms'], j['repeat_check'])"
  done
done
