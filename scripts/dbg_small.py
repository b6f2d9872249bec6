import sys, numpy as np
sys.path.insert(0, '.')
import wass_amd
from wass_amd import default_sgm_params, synth
from oracle import oracle as O
def op(p): return O.SgbmParams(p.min_disp, p.num_disp, p.win, p.P1, p.P2, p.uniq_ratio, p.disp12_max_diff, p.prefilter_cap, p.speckle_win, p.speckle_range, p.ndirs)
cases = [(64, 48, 16), (160, 120, 32), (33, 29, 16), (200, 90, 64), (320, 64, 256), (40, 300, 16), (560, 24, 512), (75, 41, 32), (700, 20, 640)]
nd = int(sys.argv[1]) if len(sys.argv) > 1 else 8
with wass_amd.Context(0) as ctx:
    for (w, h, D) in cases:
        right, left = synth.make_pair(w, h, D, frame_idx=w + h + D)
        p = default_sgm_params(D, ndirs=nd)
        print(w, h, D, 'run', flush=True)
        a = ctx.sgm_disparity(right, left, p)
        ref, st = O.dense_disparity16(right, left, op(p))
        print(w, h, D, 'mismatch', int((a != ref).sum()), flush=True)
