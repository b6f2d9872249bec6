#!/usr/bin/env python3
"""Device-side canary over every template instance of the aggregation kernels (wass_sgm_selftest): the production schedule
against one plain sweep per path, compared on the GPU.  No oracle.  WASS_GPU_LIB=<other build> python scripts/selftest.py
runs it on another build of the library (e.g. one compiled with -DWASS_REC_FENCE=0, the form that miscompiles)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import wass_amd  # noqa: E402

bad = 0
with wass_amd.Context(0) as ctx:
    for D in (64, 128, 256, 384, 512, 640, 768, 896, 1024):
        for nd in (5, 8):
            for (w, h) in ((D + 56, 40), (320 if D <= 256 else D + 64, 64), (D + 40, 17)):
                n = ctx.sgm_selftest(w, h, D, nd)
                bad += n != 0
                print(f"D={D:4d} {nd}-path {w}x{h}: {'ok' if n == 0 else str(n) + ' cells of S differ'}")
print("self-test:", "PASSED" if bad == 0 else f"{bad} configuration(s) FAILED")
sys.exit(1 if bad else 0)
