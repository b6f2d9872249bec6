#!/bin/bash
# alternate the two libraries under the C++ driver (file swap) and the harness (WASS_GPU_LIB)
cd $GRAFT_REPO_ROOT
cp wass_amd/libwassgpu.so /tmp/lib_base.so
for i in 1 2 3; do
  for v in base tri2; do
    if [ $v = base ]; then cp /tmp/lib_base.so wass_amd/libwassgpu.so; else cp wass_amd/libwassgpu.so.tri2 wass_amd/libwassgpu.so; fi
    python scripts/cxx_ab.py --rounds 1 --replicate 12 "WASS_V=$v" 2>&1 | grep steady
    python bench.py --steps 64 --warmup 8 --no-cpu-baseline --no-config-e --no-5path --no-pcie-pass --no-cxx-driver 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('harness $v', j['value'], j['ms_per_step'], j['stage_ms']['cost_volume'], j['stage_ms']['aggregate'])"
  done
done
cp /tmp/lib_base.so wass_amd/libwassgpu.so
