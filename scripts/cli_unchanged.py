#!/usr/bin/env python3
"""scripts/cli_unchanged.py [ndirs] -- bench.py's `wasscli_unchanged` record alone (4 and 8 concurrent `wass_stereo <config> <workdir>` callers over
a config-B sequence, served by the resident worker), with the server's per-call time table: for work on the per-frame client / server."""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench

debug = os.environ.get("CLI_DEBUG_IMAGES", "0") != "0"          # the reference's eight debug pictures per frame (its default)
par = int(os.environ.get("CLI_PARALLEL", "4"))                    # 1: one call after the other, what matlab/run_wass.m:242-246 does
print(json.dumps(bench.wasscli_unchanged_record(int(sys.argv[1]) if len(sys.argv) > 1 else 8, replicate=3 if debug else (4 if par == 1 else 12), parallel=par,
                                                debug_images=debug)))
