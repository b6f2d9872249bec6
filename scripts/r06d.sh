#!/bin/bash
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
( time timeout 600 python -m pytest tests/test_post_mesh_gpu.py -x -q ) > gpurun_out/r06d_pytest_mesh.log 2>&1; tail -5 gpurun_out/r06d_pytest_mesh.log
( scripts/ab_env.sh WASS_X_SKIP_ZGAP 0 1 3 ) > gpurun_out/r06d_ab_skip_zgap.log 2>&1; cat gpurun_out/r06d_ab_skip_zgap.log
( scripts/ab_bench.sh wass_amd/libwassgpu.so.oldselect 3 ) > gpurun_out/r06d_ab_zgap_select.log 2>&1; cat gpurun_out/r06d_ab_zgap_select.log
