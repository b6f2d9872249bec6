#!/usr/bin/env python3
"""Where does the aggregated volume S of the production schedule differ from the oracle's?  (development aid for the chain kernels)

    python scripts/diag_where.py W H D [ndirs] [win]

Prints the pixels (x, y) with differing cells, their position inside the blocks of k_pairx (column mod XB, row inside the K-row
segment of the split column family) and the difference per disparity for the first few."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import wass_amd  # noqa: E402
from wass_amd import synth  # noqa: E402
from oracle import oracle as O  # noqa: E402


def main():
    w, h, D = (int(a) for a in sys.argv[1:4])
    ndirs = int(sys.argv[4]) if len(sys.argv) > 4 else 8
    win = int(sys.argv[5]) if len(sys.argv) > 5 else 13
    right, left = synth.make_pair(w, h, D, frame_idx=w + h + D)
    p = wass_amd.default_sgm_params(D, ndirs=ndirs, win=win)
    with wass_amd.Context(0) as ctx:
        ctx.set_debug(True)
        ctx.sgm_disparity(right, left, p)
        Cg, Sg, rawg = ctx.sgm_debug_fetch(w, h, p)
    R = np.zeros((h, w + D), np.uint8); L = np.zeros_like(R)
    R[:, D:D + w] = right; L[:, D:D + w] = left
    op = O.SgbmParams(p.min_disp, p.num_disp, p.win, p.P1, p.P2, p.uniq_ratio, p.disp12_max_diff, p.prefilter_cap, p.speckle_win,
                      p.speckle_range, p.ndirs)
    disp, st, Co, So, rawo = O.sgbm_compute(R, L, op, dump=True)
    print("C equal:", np.array_equal(Cg, Co), " S equal:", np.array_equal(Sg, So))
    bad = np.argwhere((Sg != So).any(axis=2))
    NP = (D + 127) // 128
    XB = 10 if NP <= 3 else 8
    K = 8 if NP <= 4 else 4
    width1 = Sg.shape[1]
    mid = h // 2
    print(f"width1 {width1} h {h} mid {mid} XB {XB} K {K}; {len(bad)} pixels differ")
    for (y, x) in bad[:40]:
        seg = (y // K, y % K) if y < mid else ((h - 1 - y) // K, (h - 1 - y) % K)
        dd = (Sg[y, x].astype(int) - So[y, x].astype(int))
        nz = np.nonzero(dd)[0]
        print(f"  x={x} (col {x % XB} of block {x // XB}) y={y} ({'top' if y < mid else 'bottom'} half, seg {seg[0]} elem {seg[1]}) "
              f"{len(nz)} cells, d {nz[:4]} diff {dd[nz[:4]]}")
    if len(bad):
        ys, xs = bad[:, 0], bad[:, 1]
        print("rows:", np.unique(ys)[:30], "\ncols mod XB:", np.unique(xs % XB), " cols:", np.unique(xs)[:30])


if __name__ == "__main__":
    main()
