"""ctypes binding of libwassgpu.so (the C ABI declared in include/wass_gpu.h).

There is deliberately NO fallback: if the HIP library is missing or a call
fails, an exception is raised.  The CPU oracle under oracle/ is never used
from here.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# WASS_GPU_LIB: another build of the same library (A/B measurements); there is still no fallback of any kind
SO_PATH = os.environ.get("WASS_GPU_LIB") or os.path.join(_HERE, "libwassgpu.so")

WASS_OK = 0
WASS_ERR_COST_OVERFLOW = -5


class WassError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"libwassgpu error {code}: {msg}")
        self.code = code


class SgmParams(C.Structure):
    """wass_sgm_params -- cv::StereoSGBM set-up of wass_stereo.cpp:742-782."""
    _fields_ = [(n, C.c_int) for n in (
        "min_disp", "num_disp", "win", "P1", "P2", "uniq_ratio", "disp12_max_diff", "prefilter_cap",
        "speckle_win", "speckle_range", "ndirs", "disp_offset")] + [("dense_scale", C.c_double)]


class SgmTimings(C.Structure):
    _fields_ = [("prefilter_ms", C.c_float), ("cost_ms", C.c_float), ("aggregate_ms", C.c_float),
                ("select_ms", C.c_float), ("median_ms", C.c_float), ("total_ms", C.c_float),
                ("aggregate_launches", C.c_int), ("cost_overflow", C.c_int), ("vsum_ms", C.c_float)]


class DebugDesc(C.Structure):
    """wass_debug_desc"""
    _fields_ = [("W0", C.c_int), ("H0", C.c_int), ("roi_l", C.c_int * 4), ("roi_r", C.c_int * 4), ("d_left_crop", C.c_void_p),
                ("d_right_crop", C.c_void_p), ("d_disp16", C.c_void_p), ("d_dispf", C.c_void_p), ("num_disp", C.c_int), ("min_disp", C.c_int),
                ("disp_offset", C.c_int), ("disparity_compensation", C.c_double), ("quality", C.c_int)]


class FrameResult(C.Structure):
    """wass_frame_result"""
    _fields_ = [("zgap", C.c_double), ("n_gaps", C.c_uint64), ("component_size", C.c_uint64), ("found", C.c_int),
                ("refine_ok", C.c_int), ("ransac_plane", C.c_double * 4), ("ransac_inliers", C.c_uint64),
                ("plane", C.c_double * 4), ("refine_inliers", C.c_uint64), ("kept_after_ransac_crop", C.c_uint64),
                ("kept_final", C.c_uint64), ("n_points", C.c_uint64), ("xyzc_bytes", C.c_uint64),
                ("sgm_cost_overflow", C.c_int), ("sgm_timeout", C.c_int), ("n_triangulated", C.c_uint64),
                ("n_inliers_out", C.c_uint64), ("stage_ms", C.c_float * 5), ("reserved", C.c_int),
                ("inliers_text_bytes", C.c_uint64), ("inliers_text_unsupported", C.c_uint32), ("reserved2", C.c_uint32)]


class GridSetup(C.Structure):
    """wass_grid_setup"""
    _fields_ = [("R", C.c_double * 9), ("T", C.c_double * 3), ("baseline", C.c_double), ("xmin", C.c_double), ("xmax", C.c_double),
                ("ymin", C.c_double), ("ymax", C.c_double), ("width", C.c_int), ("height", C.c_int)]


class Geom(C.Structure):
    """wass_geom"""
    _fields_ = [("K_left", C.c_double * 9), ("K_right", C.c_double * 9), ("R", C.c_double * 9), ("T", C.c_double * 3),
                ("use_custom", C.c_int), ("R1", C.c_double * 9), ("R2", C.c_double * 9), ("P1", C.c_double * 12),
                ("P2", C.c_double * 12), ("HLi", C.c_double * 9), ("HRi", C.c_double * 9),
                ("disparity_compensation", C.c_double), ("dense_scale", C.c_double)]


class TriParams(C.Structure):
    """wass_tri_params"""
    _fields_ = [("min_angle_deg", C.c_double), ("bbox", C.c_double * 4), ("cam_distance", C.c_double)]


class RefineParams(C.Structure):
    """wass_refine_params"""
    _fields_ = [("xmin", C.c_double), ("xmax", C.c_double), ("ymin", C.c_double), ("ymax", C.c_double),
                ("max_distance", C.c_double), ("weight_by_distance", C.c_int), ("central_third_only", C.c_int)]


class PlaneResult(C.Structure):
    """wass_plane_result"""
    _fields_ = [("found", C.c_int), ("ransac_plane", C.c_double * 4), ("ransac_inliers", C.c_uint64),
                ("plane", C.c_double * 4), ("refine_inliers", C.c_uint64), ("kept_after_ransac_crop", C.c_uint64),
                ("kept_final", C.c_uint64)]


def default_sgm_params(num_disp: int, ndirs: int = 5, min_disp: int = 1, win: int = 13, p1_mult: int = 2,
                       p2_mult: int = 64, disp_offset: int = 0) -> SgmParams:
    """Defaults of SURVEY.md Appendix C (wass_stereo.cpp:742-759)."""
    return SgmParams(min_disp, num_disp, win, p1_mult * win * win, p2_mult * win * win, 1, -1, 60, -70, 16,
                     ndirs, disp_offset, 1.0)


# every symbol include/wass_gpu.h declares: name -> (restype, argtypes)
_vp, _i, _sz = C.c_void_p, C.c_int, C.c_size_t
SYMBOLS = {
    "wass_version": (C.c_char_p, []),
    "wass_ctx_create": (_i, [_i, C.POINTER(_vp)]),
    "wass_ctx_destroy": (None, [_vp]),
    "wass_last_error": (C.c_char_p, [_vp]),
    "wass_ctx_stream": (_vp, [_vp]),
    "wass_ctx_synchronize": (_i, [_vp]),
    "wass_ctx_set_debug": (_i, [_vp, _i]),
    "wass_ctx_set_tail_overlap": (_i, [_vp, _i]),
    "wass_sgm_disparity": (_i, [_vp, _vp, _vp, _i, _i, _sz, C.POINTER(SgmParams), _vp]),
    "wass_sgm_disparity_dev": (_i, [_vp, _vp, _vp, _i, _i, _sz, C.POINTER(SgmParams), _vp]),
    "wass_sgm_last_timings": (_i, [_vp, C.POINTER(SgmTimings)]),
    "wass_sgm_prev_timings": (_i, [_vp, C.POINTER(SgmTimings)]),
    "wass_sgm_call_count": (_i, [_vp, C.POINTER(C.c_uint64)]),
    "wass_sgm_call_timings": (_i, [_vp, C.c_uint64, C.POINTER(SgmTimings)]),
    "wass_sgm_debug_fetch": (_i, [_vp, _vp, _vp, _vp]),
    "wass_sgm_probe_vsum": (_i, [_vp, C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "wass_sgm_selftest": (_i, [_vp, _i, _i, _i, _i, C.POINTER(C.c_uint64)]),
    "wass_device_count": (_i, [C.POINTER(_i)]),
    "wass_ctx_set_kernel_events": (_i, [_vp, _i]),
    "wass_sgm_kernel_times": (_i, [_vp, C.c_char_p, C.c_size_t, C.POINTER(C.c_float), _i, C.POINTER(_i)]),
    "wass_disparity_postprocess": (_i, [_vp, _vp, _i, _i, C.POINTER(SgmParams), _i, _i, _i, _vp]),
    "wass_disparity_postprocess_dev": (_i, [_vp, _vp, _i, _i, C.POINTER(SgmParams), _i, _i, _i, _vp]),
    "wass_triangulate": (_i, [_vp, _vp, _i, _i, C.POINTER(_i), C.POINTER(_i), C.POINTER(Geom), _vp, _i, _i, _vp, _vp,
                              C.POINTER(TriParams), C.POINTER(_vp), C.POINTER(C.c_uint64)]),
    "wass_triangulate_dev": (_i, [_vp, _vp, _i, _i, C.POINTER(_i), C.POINTER(_i), C.POINTER(Geom), _vp, _i, _i, _vp, _vp,
                                  C.POINTER(TriParams), C.POINTER(_vp), C.POINTER(C.c_uint64)]),
    "wass_mesh_destroy": (None, [_vp]),
    "wass_mesh_size": (_i, [_vp, C.POINTER(_i), C.POINTER(_i)]),
    "wass_mesh_download": (_i, [_vp, _vp, _vp, _vp, _vp]),
    "wass_mesh_upload": (_i, [_vp, _i, _i, _vp, _vp, _vp, C.POINTER(_vp)]),
    "wass_mesh_zgap_percentile": (_i, [_vp, _vp, C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_uint64)]),
    "wass_mesh_keep_biggest_component": (_i, [_vp, _vp, C.c_double, C.POINTER(C.c_uint64)]),
    "wass_ransac_sample": (_i, [_i, _i, _i, _vp]),
    "wass_ransac_sample_seeded": (_i, [C.c_uint32, _i, _i, _i, _vp]),
    "wass_mesh_ransac_plane": (_i, [_vp, _vp, _vp, _i, C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_uint64),
                                    C.POINTER(_i)]),
    "wass_mesh_crop_plane": (_i, [_vp, _vp, C.POINTER(C.c_double), C.c_double, C.POINTER(C.c_uint64)]),
    "wass_large_gradient_mask": (_i, [_vp, _i, _i, _vp]),
    "wass_mesh_reject_codes": (_i, [_vp, _vp, _vp]),
    "wass_mesh_refine_plane": (_i, [_vp, _vp, C.POINTER(RefineParams), C.POINTER(C.c_double), C.POINTER(C.c_uint64)]),
    "wass_mesh_refinement_inliers": (_i, [_vp, _vp, C.POINTER(RefineParams), _i, C.POINTER(C.POINTER(C.c_double)), C.POINTER(C.c_uint64)]),
    "wass_mesh_remove_outliers": (_i, [_vp, _vp, C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_uint64),
                                       C.POINTER(C.c_uint64)]),
    "wass_mesh_fit_plane": (_i, [_vp, _vp, _vp, _i, C.c_double, C.POINTER(RefineParams), C.c_double,
                                 C.POINTER(PlaneResult)]),
    "wass_RT_from_plane": (None, [C.POINTER(C.c_double)] * 5),
    "wass_mesh_encode_xyzc": (_i, [_vp, _vp, C.POINTER(C.c_double), C.POINTER(_vp), C.POINTER(_sz)]),
    "wass_mesh_encode_xyzc_to": (_i, [_vp, _vp, C.POINTER(C.c_double), _vp, _sz, C.POINTER(_sz)]),
    "wass_mesh_encode_xyzc_async": (_i, [_vp, _vp, C.POINTER(C.c_double), _vp, _sz, C.POINTER(_sz)]),
    "wass_mesh_finish_frame_async": (_i, [_vp, _vp, C.c_double, _vp, _i, C.c_double, C.POINTER(RefineParams), C.c_double, _vp, _sz]),
    "wass_mesh_finish_frame_async_ex": (_i, [_vp, _vp, C.c_double, _vp, _i, C.c_double, C.POINTER(RefineParams), C.c_double, _vp, _sz,
                                             _vp, _sz, _i, _vp]),
    "wass_mesh_finish_frame_async_ex2": (_i, [_vp, _vp, C.c_double, _vp, _i, C.c_double, C.POINTER(RefineParams), C.c_double, _vp, _sz,
                                              _vp, _sz, _i, _vp, _vp, _sz]),
    "wass_format_g6": (_i, [C.c_double, _vp]),
    "wass_ctx_frame_inliers": (_i, [_vp, _vp, _sz, C.POINTER(C.c_uint64)]),
    "wass_ctx_frame_result": (_i, [_vp, C.POINTER(FrameResult)]),
    "wass_device_alloc": (_i, [_vp, _sz, C.POINTER(_vp)]),
    "wass_device_free": (None, [_vp, _vp]),
    "wass_download": (_i, [_vp, _vp, _vp, _sz]),
    "wass_download_async": (_i, [_vp, _vp, _vp, _sz]),
    "wass_resize_cubic_u8_dev": (_i, [_vp, _vp, _i, _i, _sz, _vp, _i, _i]),
    "wass_jpeg_encode_dev": (_i, [_vp, _vp, _i, _i, _i, _sz, _i, _vp, _sz, C.POINTER(_sz)]),
    "wass_debug_pictures_async": (_i, [_vp, _vp, C.POINTER(DebugDesc), _vp, C.POINTER(_sz * 8), C.POINTER(C.c_uint64)]),
    "wass_debug_picture_size": (_i, [C.POINTER(DebugDesc), _i, C.POINTER(_i), C.POINTER(_i), C.POINTER(_i)]),
    "wass_debug_pictures_result": (_i, [_vp, C.c_uint64, C.POINTER(_sz * 8)]),
    "wass_pinned_alloc": (_i, [_vp, _sz, C.POINTER(_vp)]),
    "wass_pinned_free": (None, [_vp, _vp]),
    "wass_free": (None, [_vp]),
    "wass_mesh_grid_idw": (_i, [_vp, _vp, C.POINTER(GridSetup), _vp, _vp]),
    "wass_mesh_grid_idw_ex": (_i, [_vp, _vp, C.POINTER(GridSetup), _i, _vp, _vp]),
    "wass_planes_mean_accumulate": (None, [C.POINTER(C.c_double), _i, C.POINTER(C.c_double)]),
    "wass_planes_mean_finish": (None, [C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(_i)]),
    "wass_ctx_wait_for_stream": (_i, [_vp, _vp]),
    "wass_dense_input_size": (_i, [_i, _i, C.c_double, C.POINTER(_i), C.POINTER(_i)]),
    "wass_disparity_postprocess_ex": (_i, [_vp, _vp, _i, _i, C.POINTER(SgmParams), _i, _i, _i, _i, _i, _i, _vp]),
    "wass_disparity_postprocess_ex_dev": (_i, [_vp, _vp, _i, _i, C.POINTER(SgmParams), _i, _i, _i, _i, _i, _i, _vp]),
    "wass_biggest_component_by_gradient_dev": (_i, [_vp, _vp, _i, _i, _i]),
    "wass_upload_async": (_i, [_vp, _vp, _vp, _sz]),
    "wass_burned_area_mask_dev": (_i, [_vp, _vp, _sz, _vp]),
    "wass_camera_mask_dev": (_i, [_vp, _vp, _vp, _sz, _vp]),
    "wass_clahe": (_i, [_vp, _vp, _i, _i, _sz, C.c_double, _i, _i, _vp]),
    "wass_clahe_dev": (_i, [_vp, _vp, _i, _i, _sz, C.c_double, _i, _i, _vp]),
    "wass_coll_unique_id": (_i, [_vp]),
    "wass_coll_init": (_i, [_vp, _i, _i, _vp]),
    "wass_coll_allreduce_sum_f64": (_i, [_vp, C.POINTER(C.c_double), _i]),
    "wass_stereo_rectify": (_i, [C.POINTER(C.c_double), C.POINTER(C.c_double), _i, _i, C.POINTER(C.c_double),
                                 C.POINTER(C.c_double), C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double),
                                 C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(_i), C.POINTER(_i)]),
    "wass_init_rectify_map": (_i, [C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double), _i, _i, _vp, _vp]),
    "wass_remap_cubic": (_i, [_vp, _vp, _i, _i, _sz, _vp, _vp, _i, _i, C.POINTER(_i), _vp]),
    "wass_remap_cubic_dev": (_i, [_vp, _vp, _i, _i, _sz, _vp, _vp, _i, _i, C.POINTER(_i), _vp]),
    "wass_undistort": (_i, [_vp, _vp, _i, _i, _sz, C.POINTER(C.c_double), C.POINTER(C.c_double), _i, _vp]),
    "wass_undistort_dev": (_i, [_vp, _vp, _i, _i, _sz, C.POINTER(C.c_double), C.POINTER(C.c_double), _i, _vp]),
    "wass_warp_perspective": (_i, [_vp, _vp, _i, _i, _sz, C.POINTER(C.c_double), _i, _i, C.POINTER(_i), _vp]),
    "wass_warp_perspective_dev": (_i, [_vp, _vp, _i, _i, _sz, C.POINTER(C.c_double), _i, _i, C.POINTER(_i), _vp]),
}

_lib = None


def load() -> C.CDLL:
    """Load libwassgpu.so and bind every declared symbol; raises if anything is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO_PATH):
        raise ImportError(
            f"{SO_PATH} not found: build the HIP extension first (python -m wass_amd.build). "
            "wass_amd has no CPU fallback.")
    # PyTorch bundles its own libamdhip64 (NEEDED as the unversioned "libamdhip64.so", SONAME .so.7).
    # If /opt/rocm's copy were loaded first, a later `import torch` would pull in a SECOND HIP runtime
    # and fail with "No HIP GPUs are available"; importing torch first makes both share one runtime.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(SO_PATH)
    other_build = bool(os.environ.get("WASS_GPU_LIB"))
    for name, (res, args) in SYMBOLS.items():
        if other_build and not hasattr(lib, name):
            continue                     # an OLDER build of the library in an A/B measurement: newer entry points are simply absent
        fn = getattr(lib, name)          # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib
