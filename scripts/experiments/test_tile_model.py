"""Host logic of the tile-fused aggregation schedule: the index functions the HIP kernels use
(wass_amd/csrc/tile_geom.h) drive a scalar CPU model (tests/native/tile_model.cpp) whose S volume must equal the
oracle's bit for bit -- edge-state layout, entry/store conventions, per-wave work split, partial tiles, 5 and 8 paths.
"""
import ctypes
import os
import subprocess

import numpy as np
import pytest

from wass_amd import synth

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def model():
    src = os.path.join(HERE, "tile_model.cpp")
    out_dir = os.path.join(HERE, "native", "_build")
    os.makedirs(out_dir, exist_ok=True)
    so = os.path.join(out_dir, "libtile_model.so")
    hdr = os.path.join(HERE, "tile_geom.h")
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-Wall", src, "-o", so])
    lib = ctypes.CDLL(so)
    lib.tile_model_S.restype = ctypes.c_int
    lib.tile_model_S.argtypes = [ctypes.c_void_p] + [ctypes.c_int] * 7 + [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int)]
    return lib


def _pad(right, left, D):
    h, w = right.shape
    R = np.zeros((h, w + D), np.uint8); L = np.zeros((h, w + D), np.uint8)
    R[:, D:] = right; L[:, D:] = left
    return R, L


@pytest.mark.parametrize("w,h,D,T,ndirs", [
    (40, 30, 16, 12, 8), (40, 30, 16, 12, 5),
    (37, 29, 16, 5, 8), (37, 29, 16, 4, 5),       # partial tiles on both sides
    (24, 24, 16, 12, 8),                          # exact multiple of the tile
    (9, 50, 16, 12, 8), (50, 9, 16, 12, 5),       # image smaller than one tile in one direction
    (30, 21, 32, 7, 8), (26, 17, 16, 1, 8),       # T = 1: every pixel its own tile
    (20, 14, 16, 16, 8),                          # whole image inside one partial tile
])
def test_tile_model_reproduces_oracle_S(model, oracle, w, h, D, T, ndirs):
    right, left = synth.make_pair(w, h, D, frame_idx=w * 7 + h)
    p = oracle.wass_params(D, mode=ndirs)
    R, L = _pad(right, left, D)
    disp, st, Co, So, rawo = oracle.sgbm_compute(R, L, p, dump=True)
    assert not st.overflow
    H, W1, Dd = Co.shape
    C = np.ascontiguousarray(Co, np.int16)
    S = np.zeros_like(C)
    err = ctypes.c_int(-1)
    rc = model.tile_model_S(C.ctypes.data, W1, H, Dd, p.P1, p.P2, T, ndirs, S.ctypes.data, ctypes.byref(err))
    assert rc == 0
    assert err.value == 0, "a cell was covered zero or several times by some path, or an entry state was missing"
    np.testing.assert_array_equal(S, So)
