"""Python host-side mirror of the wass_stereo dense-stereo stage.

Thin wrappers over the C ABI (include/wass_gpu.h); names follow the reference
functions they stand in for (src/wass_stereo/wass_stereo.cpp).  Host arrays are
numpy, device arrays are torch CUDA(HIP) tensors passed by raw pointer --
PyTorch is only plumbing for device memory here.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import SgmParams, SgmTimings, WassError, default_sgm_params  # noqa: F401


class Context:
    """One GPU, one stream, its scratch HBM (wass_ctx)."""

    def __init__(self, device_id: int = 0):
        self._lib = _lib.load()
        h = C.c_void_p()
        rc = self._lib.wass_ctx_create(device_id, C.byref(h))
        if rc != 0:
            raise WassError(rc, "wass_ctx_create failed (no usable GPU?)")
        self._h = h
        self.device_id = device_id

    def close(self):
        if getattr(self, "_h", None):
            self._lib.wass_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _check(self, rc: int, allow=()):
        if rc != 0 and rc not in allow:
            raise WassError(rc, self._lib.wass_last_error(self._h).decode())
        return rc

    @property
    def stream(self) -> int:
        return int(self._lib.wass_ctx_stream(self._h) or 0)

    def synchronize(self):
        self._check(self._lib.wass_ctx_synchronize(self._h))

    def set_debug(self, on: bool = True):
        """Keep intermediates that production never writes (the finished S volume) for sgm_debug_fetch."""
        self._check(self._lib.wass_ctx_set_debug(self._h, int(on)))

    # ---- sgbm_dense_stereo core (wass_stereo.cpp:820-839) ------------------
    def sgm_disparity(self, right: np.ndarray, left: np.ndarray, params: SgmParams,
                      allow_overflow: bool = False) -> np.ndarray:
        """Host arrays in, int16 fixed-point disparity (right image frame) out."""
        right = np.ascontiguousarray(right, np.uint8)
        left = np.ascontiguousarray(left, np.uint8)
        if right.shape != left.shape or right.ndim != 2:
            raise ValueError("right/left must be 2-D u8 arrays of equal shape")
        h, w = right.shape
        out = np.empty((h, w), np.int16)
        rc = self._lib.wass_sgm_disparity(self._h, right.ctypes.data, left.ctypes.data, w, h, w,
                                          C.byref(params), out.ctypes.data)
        self._check(rc, allow=(_lib.WASS_ERR_COST_OVERFLOW,) if allow_overflow else ())
        return out

    def sgm_disparity_dev(self, d_right, d_left, params: SgmParams, d_out=None):
        """torch uint8 CUDA tensors in, torch int16 CUDA tensor out (asynchronous on self.stream)."""
        import torch
        h, w = d_right.shape
        if d_out is None:
            d_out = torch.empty((h, w), dtype=torch.int16, device=d_right.device)
        rc = self._lib.wass_sgm_disparity_dev(self._h, d_right.data_ptr(), d_left.data_ptr(), w, h,
                                              d_right.stride(0), C.byref(params), d_out.data_ptr())
        self._check(rc)
        return d_out

    def sgm_timings(self) -> SgmTimings:
        t = SgmTimings()
        self._check(self._lib.wass_sgm_last_timings(self._h, C.byref(t)))
        return t

    def sgm_debug_fetch(self, w: int, h: int, params: SgmParams, want=("C", "S", "raw")):
        """Intermediates of the last sgm_disparity call (test hook)."""
        off = max(params.disp_offset, 0)
        width1 = w + off - params.min_disp
        Wp = w + params.num_disp + off
        Cv = np.empty((h, width1, params.num_disp), np.int16) if "C" in want else None
        Sv = np.empty((h, width1, params.num_disp), np.int16) if "S" in want else None
        raw = np.empty((h, Wp), np.int16) if "raw" in want else None
        self._check(self._lib.wass_sgm_debug_fetch(
            self._h, Cv.ctypes.data if Cv is not None else None, Sv.ctypes.data if Sv is not None else None,
            raw.ctypes.data if raw is not None else None))
        return Cv, Sv, raw
