#!/bin/bash
# Round 6: whole GPU suite, then the profile set (stats + FETCH/WRITE passes, B 8-path / E 8-path / B 5-path) and the bench lines on one box
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > gpurun_out/r06f_pytest_gpu.log 2>&1; tail -5 gpurun_out/r06f_pytest_gpu.log
bash scripts/r06_profiles.sh r06f 2>&1 | tail -12
