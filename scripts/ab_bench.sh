#!/bin/bash
# Alternate two builds of libwassgpu.so under bench.py (pipelined full chain) on one box: scripts/ab_bench.sh <other.so> [rounds] [bench args]
OTHER=$(readlink -f "$1"); N=${2:-3}; shift 2
for i in $(seq $N); do
  for lib in "" "$OTHER"; do
    WASS_GPU_LIB=$lib python bench.py --steps 64 --warmup 8 --no-cpu-baseline --no-cxx-driver --no-config-e --no-pcie-pass "$@" 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('${lib:+other}' or 'base ', j['value'], j['ms_per_step'], j['stage_ms'])"
  done
done
