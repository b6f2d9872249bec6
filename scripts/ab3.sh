#!/bin/bash
# scripts/ab3.sh "<lib or empty> ..." [rounds] [bench args]: alternate the shipped library ("base") and other builds under bench.py on one box
LIBS=$1; N=${2:-3}; shift 2
for i in $(seq $N); do
  for lib in base $LIBS; do
    L=""; [ "$lib" != base ] && L=$(readlink -f "$lib")
    WASS_GPU_LIB=$L python bench.py --steps 64 --warmup 8 --no-cpu-baseline --no-cxx-driver --no-config-e --no-pcie-pass --no-5path "$@" 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$(basename $lib)', j['value'], j['ms_per_step'], j['stage_ms']['cost_volume'], j['stage_ms']['aggregate'], j['roofline']['kernel_ms'])"
  done
done
