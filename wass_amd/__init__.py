"""wass_amd -- MI355X-native dense-stereo hot path of WASS's wass_stereo.

Only the pieces the hot path needs live here: csrc/ (HIP kernels + the C ABI of
include/wass_gpu.h), the ctypes binding, the Python host-side mirror of the
reference interface and the synthetic-input generator.
"""
from .stereo import (Context, Geom, Mesh, RefineParams, SgmParams, SgmTimings, TriParams, WassError,  # noqa: F401
                     RT_from_plane, default_sgm_params, init_rectify_map, make_geom, planes_mean_accumulate,
                     planes_mean_finish, ransac_sample, stereo_rectify)

__all__ = ["Context", "Geom", "Mesh", "RefineParams", "SgmParams", "SgmTimings", "TriParams", "WassError", "RT_from_plane",
           "default_sgm_params", "init_rectify_map", "make_geom", "planes_mean_accumulate", "planes_mean_finish", "ransac_sample",
           "stereo_rectify"]
