#!/bin/bash
# kernel trace of the pipelined frame with the sampled-bracket select (stats only)
mkdir -p gpurun_out/prof_r06e
export TMPDIR=/tmp
ROOT=$PWD
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_r06e/stats -o run -- python $ROOT/bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-config-e --no-cxx-driver --no-pcie-pass --no-5path > $ROOT/gpurun_out/prof_r06e/stats.log 2>&1
cd $ROOT
python scripts/summarize_prof.py gpurun_out/prof_r06e > gpurun_out/r06e_summary.txt 2>&1
head -45 gpurun_out/r06e_summary.txt
rm -rf gpurun_out/prof_r06e/stats/*/*_kernel_trace.csv
