#!/usr/bin/env python3
"""A/B of the resident worker's speculation (WASS_SERVER_SPECULATE) through the unchanged command line: alternating environments on one box."""
import json, os, sys
sys.path.insert(0, os.getcwd())
import bench
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 2):
    for spec in ("", "0"):
        if spec: os.environ["WASS_SERVER_SPECULATE"] = spec
        else: os.environ.pop("WASS_SERVER_SPECULATE", None)
        r = bench.wasscli_unchanged_record(8, replicate=12, parallel=4)
        print("SPECULATE=%s" % (spec or "default"), "4 callers", r["pairs_per_sec"], "median call", r["median_call_s"], r.get("server_ms_per_call"),
              "| 8 callers", r["parallel_8"]["pairs_per_sec"], "| debug pictures", r["with_debug_pictures"]["pairs_per_sec"],
              "| failed", r["failed_calls"] + r["parallel_8"]["failed_calls"] + r["with_debug_pictures"]["failed_calls"], flush=True)
