mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
( time timeout 400 python -m pytest tests -m gpu -x -q ) > gpurun_out/r05k_pytest_gpu.log 2>&1; tail -4 gpurun_out/r05k_pytest_gpu.log
( time timeout 330 python bench.py ) > gpurun_out/r05k_bench_default_args.json 2> gpurun_out/r05k_bench.err; tail -c 600 gpurun_out/r05k_bench.err; tail -c 300 gpurun_out/r05k_bench_default_args.json
timeout 60 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
