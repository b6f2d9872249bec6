import sys, numpy as np
sys.path.insert(0, '.')
import wass_amd
from wass_amd import default_sgm_params, synth
from oracle import oracle as O
def op(p): return O.SgbmParams(p.min_disp, p.num_disp, p.win, p.P1, p.P2, p.uniq_ratio, p.disp12_max_diff, p.prefilter_cap, p.speckle_win, p.speckle_range, p.ndirs)
w, h, D = (int(v) for v in sys.argv[1:4])
nd = int(sys.argv[4]) if len(sys.argv) > 4 else 8
with wass_amd.Context(0) as ctx:
    right, left = synth.make_pair(w, h, D, frame_idx=w + h + D)
    p = default_sgm_params(D, ndirs=nd)
    for rep in range(3):
        ctx.set_debug(True)
        got = ctx.sgm_disparity(right, left, p)
        Cg, Sg, rawg = ctx.sgm_debug_fetch(w, h, p)
        ctx.set_debug(False)
        Dp = D
        R = np.zeros((h, w + D), np.uint8); L = np.zeros((h, w + D), np.uint8)
        R[:, D:] = right; L[:, D:] = left
        disp, st, Co, So, rawo = O.sgbm_compute(R, L, op(p), dump=True)
        bad = np.argwhere(Sg != So)
        print("rep", rep, "shape", Sg.shape, "mismatch", len(bad))
        if len(bad):
            ys, xs = np.unique(bad[:, 0]), np.unique(bad[:, 1])
            print(" rows", ys[:40], "\n cols", xs[:40], "\n d", np.unique(bad[:, 2])[:20], len(np.unique(bad[:, 2])))
            pix = np.unique(bad[:, :2], axis=0)
            print(" pixels", len(pix), pix[:20].tolist())
            y, x, d = bad[0]
            print(" first", (y, x, d), int(Sg[y, x, d]), int(So[y, x, d]), "diff", (Sg[y, x].astype(int) - So[y, x].astype(int))[max(0,d-3):d+8])
            np.save("gpurun_out/bad_pix.npy", pix)
