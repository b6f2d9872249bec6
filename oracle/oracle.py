"""ctypes wrapper around the CPU oracle (oracle/libwass_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg -- never by the wass_amd product package.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libwass_oracle.so")


def build(force: bool = False) -> str:
    srcs = [os.path.join(_HERE, f) for f in ("sgbm_oracle.c", "wass_oracle.c", "rectify_oracle.c", "wass_oracle.h")]
    if force or not os.path.exists(_SO) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libwass_oracle.so"], stdout=subprocess.DEVNULL)
    return _SO


class SgbmParams(C.Structure):
    _fields_ = [(n, C.c_int) for n in (
        "min_disp", "num_disp", "block_size", "P1", "P2", "uniqueness_ratio", "disp12_max_diff",
        "prefilter_cap", "speckle_window", "speckle_range", "mode")]


class SgbmStats(C.Structure):
    _fields_ = [("max_C", C.c_int), ("max_L", C.c_int), ("overflow", C.c_int)]


class Geom(C.Structure):
    _fields_ = [("K_left", C.c_double * 9), ("K_right", C.c_double * 9),
                ("R", C.c_double * 9), ("T", C.c_double * 3), ("use_custom", C.c_int),
                ("R1", C.c_double * 9), ("R2", C.c_double * 9),
                ("P1", C.c_double * 12), ("P2", C.c_double * 12),
                ("HLi", C.c_double * 9), ("HRi", C.c_double * 9),
                ("disparity_compensation", C.c_double), ("dense_scale", C.c_double)]


class TriParams(C.Structure):
    _fields_ = [("min_angle_deg", C.c_double), ("bbox", C.c_double * 4), ("cam_distance", C.c_double)]


class RefineParams(C.Structure):
    _fields_ = [("xmin", C.c_double), ("xmax", C.c_double), ("ymin", C.c_double), ("ymax", C.c_double),
                ("max_distance", C.c_double), ("weight_by_distance", C.c_int), ("central_third_only", C.c_int)]


def wass_params(num_disp: int, mode: int = 5, min_disp: int = 1, win: int = 13,
                p1_mult: int = 2, p2_mult: int = 64) -> SgbmParams:
    """cv::StereoSGBM parameters exactly as wass_stereo.cpp:742-782 sets them."""
    return SgbmParams(min_disp, num_disp, win, p1_mult * win * win, p2_mult * win * win,
                      1, -1, 60, -70, 16, mode)


_lib = None


def lib():
    global _lib
    if _lib is None:
        # WASS_ORACLE_LIB: another build of the same sources (bench.py's cpu_baseline leg times one compiled with
        # -march=native on the box it runs on); the tests always use the portable build next to this file
        so = os.environ.get("WASS_ORACLE_LIB")
        if not so:
            build()
            so = _SO
        _lib = C.CDLL(so)
        _lib.orc_zgap_percentile.restype = C.c_double
        for f in ("orc_triangulate", "orc_keep_biggest_component", "orc_crop_plane",
                  "orc_refine_plane", "orc_encode_xyzc"):
            getattr(_lib, f).restype = C.c_size_t
    return _lib


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t)) if a is not None else None


def sgbm_compute(img1, img2, params: SgbmParams, dump=False):
    """cv::StereoSGBM::compute(img1, img2) -> int16 disparity (and optional C/S/raw dumps)."""
    img1 = np.ascontiguousarray(img1, np.uint8); img2 = np.ascontiguousarray(img2, np.uint8)
    h, w = img1.shape
    disp = np.empty((h, w), np.int16)
    st = SgbmStats()
    Cv = Sv = raw = None
    if dump:
        width1 = w - (params.min_disp + params.num_disp)
        Cv = np.empty((h, width1, params.num_disp), np.int16)
        Sv = np.empty((h, width1, params.num_disp), np.int16)
        raw = np.empty((h, w), np.int16)
    rc = lib().orc_sgbm_compute(_p(img1, C.c_uint8), _p(img2, C.c_uint8), w, h, C.byref(params),
                                _p(disp, C.c_int16), _p(Cv, C.c_int16), _p(Sv, C.c_int16),
                                _p(raw, C.c_int16), C.byref(st))
    if rc != 0:
        raise RuntimeError(f"orc_sgbm_compute failed: {rc}")
    if dump:
        return disp, st, Cv, Sv, raw
    return disp, st


def dense_disparity16(right, left, params: SgbmParams, disparity_offset: int = 0):
    right = np.ascontiguousarray(right, np.uint8); left = np.ascontiguousarray(left, np.uint8)
    h, w = right.shape
    disp = np.empty((h, w), np.int16)
    st = SgbmStats()
    rc = lib().orc_dense_disparity16(_p(right, C.c_uint8), _p(left, C.c_uint8), w, h, C.byref(params),
                                     disparity_offset, _p(disp, C.c_int16), C.byref(st))
    if rc != 0:
        raise RuntimeError(f"orc_dense_disparity16 failed: {rc}")
    return disp, st


def median3_i16(a):
    a = np.ascontiguousarray(a, np.int16)
    out = np.empty_like(a)
    lib().orc_median3_i16(_p(a, C.c_int16), _p(out, C.c_int16), a.shape[1], a.shape[0])
    return out


def clean_and_convert(d16, mindisp, num_disp, disp_offset=0, scale=1.0):
    d16 = np.ascontiguousarray(d16, np.int16)
    out = np.empty(d16.shape, np.float32)
    lib().orc_clean_and_convert(_p(d16, C.c_int16), d16.shape[1], d16.shape[0], mindisp, num_disp,
                                disp_offset, C.c_double(scale), _p(out, C.c_float))
    return out


def dilate_zero(a):
    a = np.ascontiguousarray(a, np.float32); out = np.empty_like(a)
    lib().orc_dilate_zero(_p(a, C.c_float), _p(out, C.c_float), a.shape[1], a.shape[0])
    return out


def erode_zero(a):
    a = np.ascontiguousarray(a, np.float32); out = np.empty_like(a)
    lib().orc_erode_zero(_p(a, C.c_float), _p(out, C.c_float), a.shape[1], a.shape[0])
    return out


def disparity_postprocess(d16, mindisp, num_disp, disp_offset=0, dilate_steps=1, erode_steps=2):
    d16 = np.ascontiguousarray(d16, np.int16)
    out = np.empty(d16.shape, np.float32)
    lib().orc_disparity_postprocess(_p(d16, C.c_int16), d16.shape[1], d16.shape[0], mindisp, num_disp,
                                    disp_offset, dilate_steps, erode_steps, _p(out, C.c_float))
    return out


def clahe(img, clip_limit, tiles):
    img = np.ascontiguousarray(img, np.uint8)
    out = np.empty_like(img)
    rc = lib().orc_clahe(_p(img, C.c_uint8), img.shape[1], img.shape[0], C.c_size_t(img.shape[1]), C.c_double(clip_limit), tiles, tiles,
                         _p(out, C.c_uint8))
    assert rc == 0
    return out


def resize_dsize(sw, sh, fx, fy):
    dw, dh = C.c_int(), C.c_int()
    lib().orc_resize_dsize(sw, sh, C.c_double(fx), C.c_double(fy), C.byref(dw), C.byref(dh))
    return dw.value, dh.value


def resize_cubic_u8(img, fx, fy):
    """cv::resize(img, dst, Size(), fx, fy, INTER_CUBIC) for CV_8UC1."""
    img = np.ascontiguousarray(img, np.uint8)
    sh, sw = img.shape
    dw, dh = resize_dsize(sw, sh, fx, fy)
    out = np.empty((dh, dw), np.uint8)
    lib().orc_resize_cubic_u8(_p(img, C.c_uint8), sw, sh, _p(out, C.c_uint8), dw, dh, C.c_double(1.0 / fx), C.c_double(1.0 / fy))
    return out


def resize_f32(img, ow, oh, cubic=True):
    """cv::resize(img, dst, Size(ow, oh), 0, 0, INTER_CUBIC | INTER_NEAREST) for CV_32FC1 (wass_stereo.cpp:903-904)."""
    img = np.ascontiguousarray(img, np.float32)
    sh, sw = img.shape
    out = np.empty((oh, ow), np.float32)
    fn = lib().orc_resize_cubic_f32 if cubic else lib().orc_resize_nn_f32
    fn(_p(img, C.c_float), sw, sh, _p(out, C.c_float), ow, oh, C.c_double(sw / ow), C.c_double(sh / oh))
    return out


def dense_inputs(right, left, dense_scale):
    """wass_stereo.cpp:788-796 (identity at scale 1: the reference's defect there is not reproduced)."""
    if dense_scale == 1.0:
        return right, left
    fx, fy = (dense_scale, 1.0) if dense_scale > 1.0 else (dense_scale, dense_scale)
    return resize_cubic_u8(right, fx, fy), resize_cubic_u8(left, fx, fy)


def disparity_postprocess_ex(d16, mindisp, num_disp, out_w, out_h, disp_offset=0, dense_scale=1.0, dilate_steps=1, erode_steps=2,
                             cc_threshold=0):
    d16 = np.ascontiguousarray(d16, np.int16)
    out = np.empty((out_h, out_w), np.float32)
    lib().orc_disparity_postprocess_ex(_p(d16, C.c_int16), d16.shape[1], d16.shape[0], mindisp, num_disp, disp_offset,
                                       C.c_double(dense_scale), dilate_steps, erode_steps, cc_threshold, out_w, out_h, _p(out, C.c_float))
    return out


def biggest_component_by_gradient(disp, threshold):
    out = np.ascontiguousarray(disp, np.float32).copy()
    lib().orc_biggest_component_by_gradient.restype = C.c_size_t
    area = lib().orc_biggest_component_by_gradient(_p(out, C.c_float), out.shape[1], out.shape[0], threshold)
    return out, int(area)


def filter_speckles(img, new_val, max_size, max_diff):
    out = np.ascontiguousarray(img, np.int16).copy()
    lib().orc_filter_speckles(_p(out, C.c_int16), out.shape[1], out.shape[0], new_val, max_size, max_diff)
    return out


def make_geom(g: dict, use_custom=False, disparity_compensation=0.0, dense_scale=1.0) -> Geom:
    G = Geom()
    for k in ("K_left", "K_right", "R", "T", "R1", "R2", "P1", "P2", "HLi", "HRi"):
        arr = np.asarray(g[k], np.float64).ravel()
        getattr(G, k)[:] = arr.tolist()
    G.use_custom = int(use_custom)
    G.disparity_compensation = disparity_compensation
    G.dense_scale = dense_scale
    return G


def triangulate(disp, roi_l, roi_r, geom: Geom, right_img, left_mask=None, right_mask=None,
                min_angle_deg=20.0, bbox=None, cam_distance=1.0):
    disp = np.ascontiguousarray(disp, np.float32)
    H, W = disp.shape
    right_img = np.ascontiguousarray(right_img, np.uint8)
    ih, iw = right_img.shape
    mw, mh = roi_r[2], roi_r[3]
    valid = np.zeros((mh, mw), np.uint8)
    p3d = np.zeros((mh, mw, 3), np.float64)
    gray = np.zeros((mh, mw), np.uint8)
    tp = TriParams(min_angle_deg, (C.c_double * 4)(*(bbox or (0, 0, iw, ih))), cam_distance)
    rl = (C.c_int * 4)(*roi_l); rr = (C.c_int * 4)(*roi_r)
    n = lib().orc_triangulate(_p(disp, C.c_float), W, H, rl, rr, C.byref(geom), _p(right_img, C.c_uint8),
                              iw, ih, _p(left_mask, C.c_uint8), _p(right_mask, C.c_uint8), C.byref(tp),
                              _p(valid, C.c_uint8), _p(p3d, C.c_double), _p(gray, C.c_uint8))
    return int(n), valid, p3d, gray


def triangulate_point(p, q, R, T):
    out = (C.c_double * 3)()
    lib().orc_triangulate_point((C.c_double * 2)(*p), (C.c_double * 2)(*q),
                                (C.c_double * 9)(*np.asarray(R, float).ravel()),
                                (C.c_double * 3)(*np.asarray(T, float).ravel()), out)
    return np.array(out[:])


def zgap_percentile(valid, p3d, pct):
    h, w = valid.shape
    n = C.c_size_t()
    r = lib().orc_zgap_percentile(_p(valid, C.c_uint8), _p(p3d, C.c_double), w, h, C.c_double(pct), C.byref(n))
    return float(r), int(n.value)


def keep_biggest_component(valid, p3d, zgap):
    valid = np.ascontiguousarray(valid.copy(), np.uint8)
    h, w = valid.shape
    sz = lib().orc_keep_biggest_component(_p(valid, C.c_uint8), _p(p3d, C.c_double), w, h, C.c_double(zgap))
    return valid, int(sz)


def ransac_sample(w, h, rounds, seed):
    libc = C.CDLL(None)
    libc.srand(C.c_uint(seed))
    uv = np.empty((rounds, 6), np.int32)
    lib().orc_ransac_sample(w, h, rounds, _p(uv, C.c_int32))
    return uv


def ransac_plane(valid, p3d, uv, thr):
    h, w = valid.shape
    uv = np.ascontiguousarray(uv, np.int32)
    plane = (C.c_double * 4)()
    best = C.c_size_t()
    per = np.empty(len(uv), np.int64)
    ok = lib().orc_ransac_plane(_p(valid, C.c_uint8), _p(p3d, C.c_double), w, h, _p(uv, C.c_int32), len(uv),
                                C.c_double(thr), plane, C.byref(best), _p(per, C.c_int64))
    return bool(ok), np.array(plane[:]), int(best.value), per


def crop_plane(valid, p3d, plane, thr):
    valid = np.ascontiguousarray(valid.copy(), np.uint8)
    h, w = valid.shape
    k = lib().orc_crop_plane(_p(valid, C.c_uint8), _p(p3d, C.c_double), w, h, (C.c_double * 4)(*plane), C.c_double(thr))
    return valid, int(k)


def refine_plane(valid, p3d, xmin=-9999., xmax=9999., ymin=-9999., ymax=9999., max_distance=70.0,
                 weight_by_distance=True, central_third_only=False):
    h, w = valid.shape
    rp = RefineParams(xmin, xmax, ymin, ymax, max_distance, int(weight_by_distance), int(central_third_only))
    plane = (C.c_double * 4)()
    mom = (C.c_double * 13)()
    n = lib().orc_refine_plane(_p(valid, C.c_uint8), _p(p3d, C.c_double), w, h, C.byref(rp), plane, mom)
    return np.array(plane[:]), int(n), np.array(mom[:])


def RT_from_plane(plane):
    R = (C.c_double * 9)(); T = (C.c_double * 3)(); Ri = (C.c_double * 9)(); Ti = (C.c_double * 3)()
    lib().orc_RT_from_plane((C.c_double * 4)(*plane), R, T, Ri, Ti)
    return (np.array(R[:]).reshape(3, 3), np.array(T[:]), np.array(Ri[:]).reshape(3, 3), np.array(Ti[:]))


def smallest_eigvec3(A):
    v = (C.c_double * 3)()
    lib().orc_smallest_eigvec3((C.c_double * 9)(*np.asarray(A, float).ravel()), v)
    return np.array(v[:])


def encode_xyzc(valid, p3d, plane) -> bytes:
    h, w = valid.shape
    buf = np.empty(148 + 6 * int(valid.sum()), np.uint8)
    n = lib().orc_encode_xyzc(_p(valid, C.c_uint8), _p(p3d, C.c_double), w, h, (C.c_double * 4)(*plane), _p(buf, C.c_uint8))
    return buf[:n].tobytes()


# ---- rectification, row f1 (rectify_oracle.c) ----
def inter_tab(ksize: int) -> np.ndarray:
    out = np.zeros((1024, ksize, ksize), np.int16)
    lib().orc_inter_tab(ksize, _p(out, C.c_int16))
    return out


def _d(a):
    return (C.c_double * len(np.ravel(a)))(*np.asarray(a, float).ravel())


def warp_perspective(src, H, dw, dh):
    src = np.ascontiguousarray(src, np.uint8)
    dst = np.zeros((dh, dw), np.uint8)
    lib().orc_warp_perspective(_p(src, C.c_uint8), src.shape[1], src.shape[0], C.c_size_t(src.shape[1]), _d(H), dw, dh,
                               _p(dst, C.c_uint8))
    return dst


def remap_cubic(src, map_x, map_y):
    src = np.ascontiguousarray(src, np.uint8)
    map_x = np.ascontiguousarray(map_x, np.float32); map_y = np.ascontiguousarray(map_y, np.float32)
    dh, dw = map_x.shape
    dst = np.zeros((dh, dw), np.uint8)
    lib().orc_remap_cubic(_p(src, C.c_uint8), src.shape[1], src.shape[0], C.c_size_t(src.shape[1]), _p(map_x, C.c_float),
                          _p(map_y, C.c_float), dw, dh, _p(dst, C.c_uint8))
    return dst


def init_rectify_map(K, R, P, w, h):
    mx = np.zeros((h, w), np.float32); my = np.zeros((h, w), np.float32)
    rc = lib().orc_init_rectify_map(_d(K), _d(R), _d(P), w, h, _p(mx, C.c_float), _p(my, C.c_float))
    if rc:
        raise ValueError("singular P*R")
    return mx, my


def stereo_rectify(K1, K2, w, h, R, T, alpha=1.0):
    R1 = (C.c_double * 9)(); R2 = (C.c_double * 9)(); P1 = (C.c_double * 12)(); P2 = (C.c_double * 12)()
    r1 = (C.c_int * 4)(); r2 = (C.c_int * 4)()
    rc = lib().orc_stereo_rectify(_d(K1), _d(K2), w, h, _d(R), _d(T), C.c_double(alpha), R1, R2, P1, P2, r1, r2)
    if rc:
        raise ValueError("zero baseline")
    return dict(R1=np.array(R1[:]).reshape(3, 3), R2=np.array(R2[:]).reshape(3, 3), P1=np.array(P1[:]).reshape(3, 4),
                P2=np.array(P2[:]).reshape(3, 4), roi1=tuple(r1[:]), roi2=tuple(r2[:]))


def undistort(src, K, dist):
    src = np.ascontiguousarray(src, np.uint8)
    dst = np.zeros_like(src)
    rc = lib().orc_undistort(_p(src, C.c_uint8), src.shape[1], src.shape[0], C.c_size_t(src.shape[1]), _d(K), _d(dist), len(dist),
                             _p(dst, C.c_uint8))
    if rc:
        raise ValueError("unsupported distortion model")
    return dst
