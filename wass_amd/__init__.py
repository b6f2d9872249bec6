"""wass_amd -- MI355X-native dense-stereo hot path of WASS's wass_stereo.

Only the pieces the hot path needs live here: csrc/ (HIP kernels + the C ABI of
include/wass_gpu.h), the ctypes binding, the Python host-side mirror of the
reference interface and the synthetic-input generator.
"""
from .stereo import Context, SgmParams, SgmTimings, WassError, default_sgm_params  # noqa: F401

__all__ = ["Context", "SgmParams", "SgmTimings", "WassError", "default_sgm_params"]
