#!/bin/bash
# Round 6, second GPU call: whole GPU suite, then the z-gap select A/B (six radix passes vs sampled bracket) inside the pipelined frame.
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
( time timeout 1100 python -m pytest tests -m gpu -x -q ) > gpurun_out/r06b_pytest_gpu.log 2>&1; tail -6 gpurun_out/r06b_pytest_gpu.log
( scripts/ab_bench.sh wass_amd/libwassgpu.so.oldselect 4 ) > gpurun_out/r06b_ab_zgap_select.log 2>&1; cat gpurun_out/r06b_ab_zgap_select.log
