#!/bin/bash
# Round 6, final state on one box: whole GPU suite, smoke, the driver-style bench line, the default-arguments bench line
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
TAG=${1:-r06h}
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > gpurun_out/${TAG}_pytest_gpu.log 2>&1; tail -4 gpurun_out/${TAG}_pytest_gpu.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
( time timeout 600 python bench.py --steps 20 --warmup 5 ) > gpurun_out/${TAG}_bench_driver_args.json 2> gpurun_out/${TAG}_bench_driver_args.err; tail -c 300 gpurun_out/${TAG}_bench_driver_args.err
( time timeout 900 python bench.py ) > gpurun_out/${TAG}_bench_default_args.json 2> gpurun_out/${TAG}_bench_default_args.err; tail -c 300 gpurun_out/${TAG}_bench_default_args.err
python - <<PY
import json
for f in ("gpurun_out/${TAG}_bench_driver_args.json", "gpurun_out/${TAG}_bench_default_args.json"):
    j = json.loads(open(f).read().strip().splitlines()[-1])
    r = j["roofline"]; m = j.get("mode_5path", {}); c = j.get("cxx_driver", {}); w = j.get("wasscli_unchanged", {})
    print(f, "value", j["value"], "ms", j["ms_per_step"], "frac", r["frac"], "strict", r["strict"]["frac"], "alone", r.get("sgm_stage_alone", {}).get("aggregate_ms"))
    print("  5path", m.get("pairs_per_sec"), m.get("roofline", {}).get("frac"), m.get("roofline", {}).get("strict", {}).get("frac"), "E", j.get("config_E", {}).get("roofline", {}).get("frac"))
    print("  cxx", c.get("pairs_per_sec"), c.get("host_cpu_ms_per_frame"), c.get("product_prepared_inputs"), c.get("raw_inputs"))
    print("  cli", w.get("pairs_per_sec"), w.get("with_debug_pictures", {}).get("pairs_per_sec"), w.get("one_caller_at_a_time", {}).get("pairs_per_sec"), "cpu", j.get("cpu_baseline", {}).get("value"))
PY
