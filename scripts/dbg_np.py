import sys, numpy as np
sys.path.insert(0, '.')
import wass_amd
from wass_amd import default_sgm_params, synth
from oracle import oracle as O
def op(p): return O.SgbmParams(p.min_disp, p.num_disp, p.win, p.P1, p.P2, p.uniq_ratio, p.disp12_max_diff, p.prefilter_cap, p.speckle_win, p.speckle_range, p.ndirs)
D = int(sys.argv[1])
with wass_amd.Context(0) as ctx:
    for (w, h) in ((1400, 700), (700, 1500)):
        right, left = synth.make_pair(w, h, D, frame_idx=5)
        p = default_sgm_params(D, ndirs=8)
        ref, st = O.dense_disparity16(right, left, op(p))
        res = [ctx.sgm_disparity(right, left, p) for _ in range(4)]
        print(w, h, D, 'mismatch vs oracle per run', [int((r != ref).sum()) for r in res], flush=True)
