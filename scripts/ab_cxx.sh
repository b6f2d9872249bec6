#!/bin/bash
# Alternate two environments under the shipped C++ sequence driver (bench.py's cxx_driver leg: 192 config-B workdirs, one worker):
#   scripts/ab_cxx.sh "VAR=a" "VAR=b" [rounds]
A=$1; B=$2; N=${3:-3}
for i in $(seq $N); do
  for e in "$A" "$B"; do
    env $e python -c "
import bench, json
r = bench.cxx_driver_record(8)
print('$e', r.get('pairs_per_sec'), 'pairs/s steady,', r.get('pairs_per_sec_incl_startup'), 'incl. start-up,', r.get('host_cpu_ms_per_frame'), 'ms CPU/frame,', r.get('host_cores_busy'), 'cores', r.get('error', ''))" 2>/dev/null
  done
done
