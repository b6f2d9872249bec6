"""sgbm_direct.py -- a SECOND, independent restatement of cv::StereoSGBM (OpenCV 4.5.5, MODE_SGBM / MODE_HH).

TEST INFRASTRUCTURE ONLY (see oracle/wass_oracle.h).  PARITY UNPINNED: OpenCV is not available in this image; this file
is written from SURVEY.md Appendix A alone -- direct-form definitions, no sliding windows, no buffers reused between
paths, no code shared with oracle/sgbm_oracle.c -- so that a misreading of the appendix in one implementation does not
silently pass as "parity" in the other (tests/test_oracle_direct.py compares the two on C, S, the raw disparity and
the final map).  Pure numpy + Python loops: small images only.

Reference call sites: src/wass_stereo/wass_stereo.cpp:775-782 (parameters), :837 (compute(right, left)).
"""
from __future__ import annotations

import numpy as np

MAX_COST = 32767          # SHRT_MAX (A.1)
DISP_SHIFT = 4


def derived(min_disp, num_disp, block_size, P1, P2, uniqueness_ratio, disp12_max_diff, prefilter_cap):
    """A.1: parameters as OpenCV derives them."""
    d = {}
    d["minD"] = min_disp
    d["D"] = num_disp
    d["maxD"] = min_disp + num_disp
    d["SW2"] = d["SH2"] = block_size // 2
    d["ftzero"] = max(prefilter_cap, 15) | 1
    d["uniq"] = uniqueness_ratio if uniqueness_ratio >= 0 else 10
    d["d12"] = disp12_max_diff if disp12_max_diff > 0 else 1        # Appendix F: <= 0 does NOT disable the check
    d["P1"] = P1 if P1 > 0 else 2
    d["P2"] = max(P2 if P2 > 0 else 5, d["P1"] + 1)
    d["INVALID"] = (min_disp - 1) << DISP_SHIFT
    return d


def prefilter(img, ftzero):
    """A.2 step 1: clipped x-Sobel channel and raw channel; columns 0 and W-1 of BOTH are tab[0] = ftzero."""
    I = img.astype(np.int64)
    h, w = I.shape
    up = np.vstack([I[:1], I[:-1]])          # row y-1, replicated at the top
    dn = np.vstack([I[1:], I[-1:]])          # row y+1, replicated at the bottom
    sob = np.full((h, w), ftzero, np.int64)
    raw = np.full((h, w), ftzero, np.int64)
    g = 2 * (I[:, 2:] - I[:, :-2]) + (up[:, 2:] - up[:, :-2]) + (dn[:, 2:] - dn[:, :-2])
    sob[:, 1:-1] = np.clip(g, -ftzero, ftzero) + ftzero
    raw[:, 1:-1] = I[:, 1:-1]
    return sob, raw


def _interval(ch):
    """A.2 step 2: per pixel the min / max over the value and its two half-pixel neighbours (integer division, the
    neighbour replaced by the pixel itself outside the row)."""
    left = np.hstack([ch[:, :1], ch[:, :-1]])
    right = np.hstack([ch[:, 1:], ch[:, -1:]])
    a = (ch + left) // 2
    b = (ch + right) // 2
    return np.minimum(np.minimum(a, b), ch), np.maximum(np.maximum(a, b), ch)


def pixel_cost(img1, img2, p):
    """A.2: pix[y][x][d] for x in [0, width1), d in [0, D): BT(Sobel) + (BT(raw) >> 2)."""
    h, w = img1.shape
    D, minD, maxD = p["D"], p["minD"], p["maxD"]
    width1 = w - maxD
    out = np.zeros((h, width1, D), np.int64)
    ch1 = prefilter(img1, p["ftzero"])
    ch2 = prefilter(img2, p["ftzero"])
    for c, shift in ((0, 0), (1, 2)):
        u = ch1[c]; v = ch2[c]
        u0, u1 = _interval(u)
        v0, v1 = _interval(v)
        for d in range(D):
            X = np.arange(width1) + maxD                 # image-1 column of index x (minX1 = maxD)
            X2 = X - (d + minD)                          # image-2 column
            uu, uu0, uu1 = u[:, X], u0[:, X], u1[:, X]
            vv, vv0, vv1 = v[:, X2], v0[:, X2], v1[:, X2]
            c0 = np.maximum(0, np.maximum(uu - vv1, vv0 - uu))
            c1 = np.maximum(0, np.maximum(vv - uu1, uu0 - vv))
            out[:, :, d] += np.minimum(c0, c1) >> shift
    return out


def block_cost(pix, p):
    """A.3 (direct double sum with replicate clamps, the horizontal one in the width1 domain); returned WITHOUT the
    +P2 bias, like the oracle's dump."""
    h, width1, D = pix.shape
    SW2, SH2 = p["SW2"], p["SH2"]
    hs = np.zeros_like(pix)
    for i in range(-SW2, SW2 + 1):
        hs += pix[:, np.clip(np.arange(width1) + i, 0, width1 - 1), :]
    C = np.zeros_like(pix)
    for j in range(-SH2, SH2 + 1):
        C += hs[np.clip(np.arange(h) + j, 0, h - 1)]
    return C


def path_costs(C, p, r):
    """A.4 for ONE path with predecessor offset r = (rx, ry): L_r[y][x][d].  Out-of-image predecessors are the
    all-zero state; sentinels MAX_COST at d = -1 and d = D."""
    h, width1, D = C.shape
    P1, P2 = p["P1"], p["P2"]
    rx, ry = r
    L = np.zeros((h, width1, D), np.int64)
    ys = range(h) if ry <= 0 else range(h - 1, -1, -1)
    xs = list(range(width1)) if rx <= 0 else list(range(width1 - 1, -1, -1))
    for y in ys:
        for x in xs:
            px, py = x + rx, y + ry
            Lp = L[py, px] if (0 <= px < width1 and 0 <= py < h) else np.zeros(D, np.int64)
            mn = int(Lp.min())
            ext = np.concatenate([[MAX_COST], Lp, [MAX_COST]])
            best = np.minimum(np.minimum(Lp, ext[:-2] + P1), np.minimum(ext[2:] + P1, mn + P2))
            L[y, x] = C[y, x] + best - mn                 # the +P2 bias of C and the -P2 inside delta cancel
    return L


PATHS_SGBM = [(-1, 0), (-1, -1), (0, -1), (1, -1), (1, 0)]       # predecessor offsets, MODE_SGBM
PATHS_HH = PATHS_SGBM + [(1, 1), (0, 1), (-1, 1)]                # MODE_HH adds the three up-going ones


def aggregate(C, p, mode):
    """S = sat16(sum over the paths of L_r); also returns the largest path cost seen (A.7)."""
    S = np.zeros(C.shape, np.int64)
    maxL = 0
    for r in (PATHS_HH if mode == 8 else PATHS_SGBM):
        L = path_costs(C, p, r)
        maxL = max(maxL, int(L.max()))
        S += L
    return np.minimum(S, MAX_COST), maxL


def select_row(Srow, w, p):
    """A.5 for one row: S[width1][D] -> disp1[w] (int16 values)."""
    width1, D = Srow.shape
    minD, maxD = p["minD"], p["maxD"]
    INV = p["INVALID"]
    disp1 = np.full(w, INV, np.int64)
    disp2 = np.full(w, INV, np.int64)
    disp2cost = np.full(w, MAX_COST, np.int64)
    minX1 = maxD
    for x in range(width1 - 1, -1, -1):
        Sp = Srow[x]
        minS, best = MAX_COST, -1
        for d in range(D):                        # first minimum, strict <
            if Sp[d] < minS:
                minS, best = int(Sp[d]), d
        unique = True
        for d in range(D):
            if Sp[d] * (100 - p["uniq"]) < minS * 100 and abs(best - d) > 1:
                unique = False
                break
        if not unique:
            continue
        X2 = x + minX1 - best - minD
        if disp2cost[X2] > minS:
            disp2cost[X2] = minS
            disp2[X2] = best + minD
        if 0 < best < D - 1:
            denom2 = max(int(Sp[best - 1] + Sp[best + 1] - 2 * Sp[best]), 1)
            num = int(Sp[best - 1] - Sp[best + 1]) * 16 + denom2
            q = abs(num) // (denom2 * 2)
            d16 = best * 16 + (q if num >= 0 else -q)            # C division truncates toward zero
        else:
            d16 = best * 16
        disp1[x + minX1] = d16 + minD * 16
    for X in range(minX1, w):
        d1 = int(disp1[X])
        if d1 == INV:
            continue
        _d, d_ = d1 >> 4, (d1 + 15) >> 4
        _x, x_ = X - _d, X - d_
        bad0 = 0 <= _x < w and disp2[_x] >= minD and abs(int(disp2[_x]) - _d) > p["d12"]
        bad1 = 0 <= x_ < w and disp2[x_] >= minD and abs(int(disp2[x_]) - d_) > p["d12"]
        if bad0 and bad1:
            disp1[X] = INV
    return disp1


def median3(a):
    """A.6: 3x3 median with replicated border."""
    h, w = a.shape
    pad = np.pad(a, 1, mode="edge")
    stack = np.stack([pad[i:i + h, j:j + w] for i in range(3) for j in range(3)])
    return np.sort(stack, axis=0)[4]


def compute(img1, img2, min_disp, num_disp, block_size, P1, P2, uniqueness_ratio=1, disp12_max_diff=-1, prefilter_cap=60,
            mode=5):
    """cv::StereoSGBM::compute(img1, img2): returns (disp16 int16 [h][w], C, S, raw)."""
    p = derived(min_disp, num_disp, block_size, P1, P2, uniqueness_ratio, disp12_max_diff, prefilter_cap)
    h, w = img1.shape
    C = block_cost(pixel_cost(img1, img2, p), p)
    S, maxL = aggregate(C, p, mode)
    assert C.max() + p["P2"] <= MAX_COST and maxL <= MAX_COST, "outside the int16 range of A.7"
    raw = np.stack([select_row(S[y], w, p) for y in range(h)])
    return median3(raw).astype(np.int16), C.astype(np.int16), S.astype(np.int16), raw.astype(np.int16)
