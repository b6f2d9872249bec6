#!/bin/bash
# Round 6, first GPU call: the SGM parity tests on the new 5-path schedule, then the driver-style bench line.
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
( time timeout 600 python -m pytest tests/test_sgm_gpu.py tests/test_fullsize_gpu.py -x -q ) > gpurun_out/r06a_pytest_sgm.log 2>&1; tail -6 gpurun_out/r06a_pytest_sgm.log
( time timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline ) > gpurun_out/r06a_bench_driver_args.json 2> gpurun_out/r06a_bench.err; tail -c 400 gpurun_out/r06a_bench.err
python - <<'PY'
import json
j = json.loads(open("gpurun_out/r06a_bench_driver_args.json").read().strip().splitlines()[-1])
print("value", j["value"], "ms", j["ms_per_step"], "stage", j["stage_ms"])
print("roofline", {k: j["roofline"][k] for k in ("frac", "ms", "kernel_ms")}, "strict", j["roofline"]["strict"])
m = j.get("mode_5path", {})
print("5path", m.get("pairs_per_sec"), m.get("ms_per_step"), m.get("stage_ms"), m.get("roofline", {}).get("frac"), m.get("roofline", {}).get("strict"), m.get("roofline", {}).get("kernel_ms"))
print("alone", j["roofline"].get("sgm_stage_alone"))
PY
