#!/usr/bin/env python3
"""scripts/fuzz_sgm.py [n] [seed] -- random shapes / disparity ranges / windows / penalties through the SGM stage in both path modes against the
oracle: final map and the aggregated volume S, bit for bit.  A wider net than the suite's fixed cases, for after a change of schedule
(round 6: the 5-path mode on the fused kernel).  Needs an MI355X."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import wass_amd                                  # noqa: E402
from oracle import oracle as O                   # noqa: E402
from wass_amd import default_sgm_params, synth   # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 120
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 6)
bad = done = skipped = 0
with wass_amd.Context(0) as ctx:
    ctx.set_debug(True)
    for it in range(n):
        D = int(rng.choice([16, 32, 48, 64, 96, 128, 160, 256, 272, 384, 400, 512, 640]))
        h = int(rng.choice([1, 2, 3, 7, 8, 9, 15, 16, 17, 31, 33, 40, 63, 64, 65, 90, 130]))
        w = int(rng.integers(12, 140)) + (D // 3)
        win = int(rng.choice([3, 5, 7, 9, 11, 13]))
        ndirs = int(rng.choice([5, 5, 8]))
        mind = int(rng.integers(0, 4))
        off = int(rng.choice([0, 0, 0, 3, -4])) if D >= 32 else 0
        if rng.random() < 0.8:
            right, left = synth.make_pair(w, h, D, frame_idx=int(rng.integers(0, 10 ** 6)))
        else:
            right = rng.integers(0, 256, (h, w), dtype=np.uint8)
            left = np.roll(right, int(rng.integers(1, 9)), axis=1)
        p = default_sgm_params(D, ndirs=ndirs, win=win, min_disp=mind, disp_offset=off, p2_mult=int(rng.choice([32, 16, 8])))
        p.uniq_ratio = int(rng.choice([0, 5, 10, 15]))
        op = O.SgbmParams(p.min_disp, p.num_disp, p.win, p.P1, p.P2, p.uniq_ratio, p.disp12_max_diff, p.prefilter_cap, p.speckle_win, p.speckle_range, p.ndirs)
        offp, comp = max(off, 0), max(-off, 0)
        Wp = w + D + offp
        R = np.zeros((h, Wp), np.uint8); L = np.zeros((h, Wp), np.uint8)
        R[:, D:D + w] = right
        L[:, D + offp - comp:D + offp - comp + w] = left
        disp, st, Co, So, rawo = O.sgbm_compute(R, L, op, dump=True)
        if st.overflow:
            skipped += 1
            continue
        got = ctx.sgm_disparity(right, left, p)
        Cg, Sg, rawg = ctx.sgm_debug_fetch(w, h, p)
        ok = np.array_equal(Sg, So) and np.array_equal(got, disp[:, D:D + w]) and np.array_equal(Cg, Co)
        done += 1
        if not ok:
            bad += 1
            print(f"MISMATCH: w={w} h={h} D={D} win={win} ndirs={ndirs} minD={mind} off={off} P2={p.P2} uniq={p.uniq_ratio}: "
                  f"C {int((Cg != Co).sum())} S {int((Sg != So).sum())} map {int((got != disp[:, D:D + w]).sum())}", flush=True)
print(f"fuzz_sgm: {done} cases compared ({skipped} outside the int16 range skipped), {bad} mismatches")
sys.exit(1 if bad else 0)
