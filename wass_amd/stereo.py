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
from ._lib import (Geom, PlaneResult, RefineParams, SgmParams, SgmTimings, TriParams, WassError,  # noqa: F401
                   default_sgm_params)


def dense_input_size(w: int, h: int, dense_scale: float):
    """Size of the SGBM inputs / of the raw disparity map for a DENSE_SCALE (wass_stereo.cpp:788-796)."""
    ws, hs = C.c_int(), C.c_int()
    if _lib.load().wass_dense_input_size(w, h, C.c_double(dense_scale), C.byref(ws), C.byref(hs)) != 0:
        raise ValueError("bad DENSE_SCALE")
    return ws.value, hs.value


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

    def set_tail_overlap(self, on: bool = True):
        """Run everything after the SGM call on a second stream (see wass_ctx_set_tail_overlap)."""
        self._check(self._lib.wass_ctx_set_tail_overlap(self._h, int(on)))

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
        ws, hs = dense_input_size(w, h, params.dense_scale)      # DENSE_SCALE != 1: the map has the size of the resized crops
        out = np.empty((hs, ws), np.int16)
        rc = self._lib.wass_sgm_disparity(self._h, right.ctypes.data, left.ctypes.data, w, h, w,
                                          C.byref(params), out.ctypes.data)
        self._check(rc, allow=(_lib.WASS_ERR_COST_OVERFLOW,) if allow_overflow else ())
        return out

    def sgm_disparity_dev(self, d_right, d_left, params: SgmParams, d_out=None):
        """torch uint8 CUDA tensors in, torch int16 CUDA tensor out (asynchronous on self.stream)."""
        import torch
        for name, t in (("d_right", d_right), ("d_left", d_left)):
            if t.dtype != torch.uint8 or not t.is_cuda or t.dim() != 2 or t.stride(1) != 1:
                raise ValueError(f"{name}: expected a 2-D uint8 CUDA tensor with unit inner stride")
            if t.device.index != self.device_id:
                raise ValueError(f"{name} lives on {t.device}, the context on cuda:{self.device_id}")
        if d_left.shape != d_right.shape or d_left.stride(0) != d_right.stride(0):
            raise ValueError("d_right and d_left must have the same shape and pitch")
        h, w = d_right.shape
        # DENSE_SCALE != 1: the library resizes the crops first and writes the map at THAT size (wass_dense_input_size)
        ws, hs = dense_input_size(w, h, params.dense_scale) if params.dense_scale != 1.0 else (w, h)
        if d_out is None:
            d_out = torch.empty((hs, ws), dtype=torch.int16, device=d_right.device)
        elif d_out.dtype != torch.int16 or not d_out.is_contiguous() or tuple(d_out.shape) != (hs, ws):
            raise ValueError(f"d_out: expected a contiguous int16 tensor of shape ({hs}, {ws}) (the size of the resized inputs)")
        rc = self._lib.wass_sgm_disparity_dev(self._h, d_right.data_ptr(), d_left.data_ptr(), w, h,
                                              d_right.stride(0), C.byref(params), d_out.data_ptr())
        self._check(rc)
        return d_out

    def wait_for_stream(self, stream_handle: int):
        """Order everything enqueued on this context from now on after the work already enqueued on another HIP
        stream (e.g. torch.cuda.current_stream().cuda_stream, which produced the input tensors)."""
        self._check(self._lib.wass_ctx_wait_for_stream(self._h, C.c_void_p(stream_handle)))

    def upload_async(self, d_dst, h_src):
        """Pinned host tensor -> device tensor on the context's SGM stream (wass_upload_async)."""
        n = h_src.numel() * h_src.element_size()
        if d_dst.numel() * d_dst.element_size() != n or not d_dst.is_contiguous() or not h_src.is_contiguous():
            raise ValueError("upload_async: contiguous tensors of equal size expected")
        self._check(self._lib.wass_upload_async(self._h, d_dst.data_ptr(), h_src.data_ptr(), n))

    def burned_area_mask_dev(self, d_img, d_mask):
        """d_mask = d_img <= 254 on the context's SGM stream (DISCARD_BURNED_AREAS, wass_stereo.cpp:1072)."""
        self._check(self._lib.wass_burned_area_mask_dev(self._h, d_img.data_ptr(), d_img.numel(), d_mask.data_ptr()))

    def sgm_selftest(self, w: int, h: int, num_disp: int, ndirs: int = 8) -> int:
        """Device-side canary (wass_sgm_selftest): number of cells of S in which the production schedule and the plain
        per-path sweeps disagree (0 = fine); no oracle involved."""
        n = C.c_uint64()
        rc = self._lib.wass_sgm_selftest(self._h, w, h, num_disp, ndirs, C.byref(n))
        if rc != 0 and n.value in (0, 2 ** 64 - 1):
            self._check(rc)
        return int(n.value)

    def sgm_probe_vsum(self):
        """(plain_ms, production_ms) of the vertical block sum on the last call's horizontal sums (wass_sgm_probe_vsum)."""
        a, b = C.c_float(), C.c_float()
        self._check(self._lib.wass_sgm_probe_vsum(self._h, C.byref(a), C.byref(b)))
        return float(a.value), float(b.value)

    def set_kernel_events(self, on: bool) -> None:
        """Bracket every launch of the cost stage and of the aggregation family with hipEvents (measurement only)."""
        self._check(self._lib.wass_ctx_set_kernel_events(self._h, 1 if on else 0))

    def sgm_kernel_times(self):
        """[(kernel name, ms)] of the last SGM call's launches, in launch order (needs set_kernel_events(True) before the call)."""
        names = C.create_string_buffer(4096)
        ms = (C.c_float * 32)()
        n = C.c_int()
        self._check(self._lib.wass_sgm_kernel_times(self._h, names, 4096, ms, 32, C.byref(n)))
        return list(zip(names.value.decode().split("\n"), [float(ms[i]) for i in range(n.value)]))

    def sgm_timings(self, previous: bool = False) -> SgmTimings:
        """Stage times of the last SGM call (previous=True: of the call before it, which a pipelined driver can
        read without waiting for the frame it has just enqueued)."""
        t = SgmTimings()
        f = self._lib.wass_sgm_prev_timings if previous else self._lib.wass_sgm_last_timings
        self._check(f(self._h, C.byref(t)))
        return t

    def sgm_call_count(self) -> int:
        """SGM calls of this context that were enqueued completely (the 1-based number of the last one)."""
        n = C.c_uint64()
        self._check(self._lib.wass_sgm_call_count(self._h, C.byref(n)))
        return int(n.value)

    def sgm_call_timings(self, call: int) -> SgmTimings:
        """Stage times of SGM call number `call` (1-based; the last four calls are kept): a driver that runs two frames ahead reads
        call n's after call n+2 has been enqueued, without waiting for anything that is still running."""
        t = SgmTimings()
        self._check(self._lib.wass_sgm_call_timings(self._h, int(call), C.byref(t)))
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

    # ---- disparity clean-up (wass_stereo.cpp:853-945) -----------------------
    def disparity_postprocess(self, disp16: np.ndarray, params: SgmParams, dilate_steps: int = 1,
                              erode_steps: int = 2, median_wsize: int = 0) -> np.ndarray:
        disp16 = np.ascontiguousarray(disp16, np.int16)
        h, w = disp16.shape
        out = np.empty((h, w), np.float32)
        self._check(self._lib.wass_disparity_postprocess(self._h, disp16.ctypes.data, w, h, C.byref(params),
                                                         dilate_steps, erode_steps, median_wsize, out.ctypes.data))
        return out

    def disparity_postprocess_ex(self, disp16: np.ndarray, params: SgmParams, out_w: int, out_h: int, dilate_steps: int = 1,
                                 erode_steps: int = 2, median_wsize: int = 0, cc_threshold: int = 0) -> np.ndarray:
        """wass_stereo.cpp:853-986 with every option (DENSE_SCALE, biggest component by gradient)."""
        disp16 = np.ascontiguousarray(disp16, np.int16)
        hs, ws = disp16.shape
        out = np.empty((out_h, out_w), np.float32)
        self._check(self._lib.wass_disparity_postprocess_ex(self._h, disp16.ctypes.data, ws, hs, C.byref(params), dilate_steps,
                                                            erode_steps, median_wsize, cc_threshold, out_w, out_h, out.ctypes.data))
        return out

    def disparity_postprocess_dev(self, d_disp16, params: SgmParams, dilate_steps: int = 1, erode_steps: int = 2,
                                  median_wsize: int = 0, d_out=None):
        import torch
        h, w = d_disp16.shape
        if d_out is None:
            d_out = torch.empty((h, w), dtype=torch.float32, device=d_disp16.device)
        self._check(self._lib.wass_disparity_postprocess_dev(self._h, d_disp16.data_ptr(), w, h, C.byref(params),
                                                             dilate_steps, erode_steps, median_wsize, d_out.data_ptr()))
        return d_out

    def frame_result(self):
        """Wait for the frame enqueued by Mesh.finish_frame_async and return its wass_frame_result."""
        res = _lib.FrameResult()
        self._check(self._lib.wass_ctx_frame_result(self._h, C.byref(res)))
        return res

    def debug_pictures_result(self, ticket: int):
        """Waits for the pictures of that ticket; the eight file sizes (0: did not fit into its slot, not written)."""
        n = (C.c_size_t * 8)()
        self._check(self._lib.wass_debug_pictures_result(self._h, int(ticket), C.byref(n)))
        return [int(v) for v in n]

    def jpeg_encode(self, d_img, quality: int = 95) -> bytes:
        """A picture resident in HBM (torch uint8 CUDA tensor, H x W grey or H x W x 3 r,g,b) as a complete baseline JPEG file
        (wass_jpeg_encode_dev: the encoder of the device-side debug pictures, csrc/jpeg.hip)."""
        import torch
        if d_img.dtype != torch.uint8 or not d_img.is_cuda or d_img.dim() not in (2, 3) or not d_img.is_contiguous():
            raise ValueError("expected a contiguous uint8 CUDA tensor, H x W or H x W x 3")
        h, w = d_img.shape[:2]
        ch = 1 if d_img.dim() == 2 else d_img.shape[2]
        cap = w * h * ch * 2 + 4096
        out = np.empty(cap, np.uint8)
        n = C.c_size_t()
        self.wait_for_stream(torch.cuda.current_stream(self.device_id).cuda_stream)
        self._check(self._lib.wass_jpeg_encode_dev(self._h, d_img.data_ptr(), w, h, ch, w * ch, quality, out.ctypes.data, cap, C.byref(n)))
        return out[:n.value].tobytes()

    def frame_inliers(self, n_points: int) -> np.ndarray:
        """The selected inlier points of the frame whose result was read last ((n, 3) float64), fetched on demand: the text-only form
        of the frame tail leaves them on the device.  Valid until the next finish_frame_async of this context."""
        out = np.empty((max(int(n_points), 1), 3), np.float64)
        n = C.c_uint64()
        self._check(self._lib.wass_ctx_frame_inliers(self._h, out.ctypes.data, out.shape[0], C.byref(n)))
        return out[:int(n.value)]

    # ---- rectification resamplers (wass_stereo.cpp:515-516, 603-604) ----------
    @staticmethod
    def _roi(roi):
        return (C.c_int * 4)(*roi) if roi is not None else None

    def remap_cubic(self, src: np.ndarray, map_x: np.ndarray, map_y: np.ndarray, roi=None) -> np.ndarray:
        """cv::remap(src, dst, map_x, map_y, INTER_CUBIC); roi = (x, y, w, h) returns only that window."""
        src = np.ascontiguousarray(src, np.uint8)
        map_x = np.ascontiguousarray(map_x, np.float32); map_y = np.ascontiguousarray(map_y, np.float32)
        dh, dw = map_x.shape
        out = np.empty((roi[3], roi[2]) if roi is not None else (dh, dw), np.uint8)
        self._check(self._lib.wass_remap_cubic(self._h, src.ctypes.data, src.shape[1], src.shape[0], src.shape[1],
                                               map_x.ctypes.data, map_y.ctypes.data, dw, dh, self._roi(roi), out.ctypes.data))
        return out

    def remap_cubic_dev(self, d_src, d_map_x, d_map_y, roi=None, d_out=None):
        import torch
        dh, dw = d_map_x.shape
        if d_out is None:
            d_out = torch.empty((roi[3], roi[2]) if roi is not None else (dh, dw), dtype=torch.uint8, device=d_src.device)
        self._check(self._lib.wass_remap_cubic_dev(self._h, d_src.data_ptr(), d_src.shape[1], d_src.shape[0], d_src.stride(0),
                                                   d_map_x.data_ptr(), d_map_y.data_ptr(), dw, dh, self._roi(roi),
                                                   d_out.data_ptr()))
        return d_out

    def undistort(self, src: np.ndarray, K, dist) -> np.ndarray:
        """cv::undistort(src, dst, K, dist) of wass_prepare (wass_prepare.cpp:268)."""
        src = np.ascontiguousarray(src, np.uint8)
        out = np.empty_like(src)
        Kc = (C.c_double * 9)(*np.asarray(K, float).ravel())
        dist = np.asarray(dist, float).ravel()
        dc = (C.c_double * max(len(dist), 1))(*dist)
        self._check(self._lib.wass_undistort(self._h, src.ctypes.data, src.shape[1], src.shape[0], src.shape[1], Kc, dc, len(dist),
                                             out.ctypes.data))
        return out

    def clahe(self, src: np.ndarray, clip_limit: float, tiles: int) -> np.ndarray:
        """cv::createCLAHE(clip_limit, Size(tiles, tiles))->apply(src, dst) of wass_prepare (wass_prepare.cpp:257-262,446-449)."""
        src = np.ascontiguousarray(src, np.uint8)
        out = np.empty_like(src)
        self._check(self._lib.wass_clahe(self._h, src.ctypes.data, src.shape[1], src.shape[0], src.shape[1], float(clip_limit), int(tiles),
                                         int(tiles), out.ctypes.data))
        return out

    def warp_perspective(self, src: np.ndarray, H, dw: int, dh: int, roi=None) -> np.ndarray:
        """cv::warpPerspective(src, dst, H, Size(dw, dh)) (INTER_LINEAR, zero border)."""
        src = np.ascontiguousarray(src, np.uint8)
        out = np.empty((roi[3], roi[2]) if roi is not None else (dh, dw), np.uint8)
        Hc = (C.c_double * 9)(*np.asarray(H, float).ravel())
        self._check(self._lib.wass_warp_perspective(self._h, src.ctypes.data, src.shape[1], src.shape[0], src.shape[1], Hc,
                                                    dw, dh, self._roi(roi), out.ctypes.data))
        return out

    def warp_perspective_dev(self, d_src, H, dw: int, dh: int, roi=None, d_out=None):
        import torch
        if d_out is None:
            d_out = torch.empty((roi[3], roi[2]) if roi is not None else (dh, dw), dtype=torch.uint8, device=d_src.device)
        Hc = (C.c_double * 9)(*np.asarray(H, float).ravel())
        self._check(self._lib.wass_warp_perspective_dev(self._h, d_src.data_ptr(), d_src.shape[1], d_src.shape[0],
                                                        d_src.stride(0), Hc, dw, dh, self._roi(roi), d_out.data_ptr()))
        return d_out

    # ---- triangulate(StereoMatchEnv&) (wass_stereo.cpp:1039-1386) ------------
    def triangulate(self, disp_roi, W, H, roi_l, roi_r, geom: Geom, right_img, left_mask=None, right_mask=None,
                    min_angle_deg=20.0, bbox=None, cam_distance=1.0):
        """Host arrays in -> (Mesh, n_points)."""
        disp_roi = np.ascontiguousarray(disp_roi, np.float32)
        right_img = np.ascontiguousarray(right_img, np.uint8)
        ih, iw = right_img.shape
        assert disp_roi.shape == (roi_r[3], roi_r[2])
        lm = np.ascontiguousarray(left_mask, np.uint8) if left_mask is not None else None
        rm = np.ascontiguousarray(right_mask, np.uint8) if right_mask is not None else None
        tp = TriParams(min_angle_deg, (C.c_double * 4)(*(bbox or (0, 0, iw, ih))), cam_distance)
        h = C.c_void_p(); n = C.c_uint64()
        self._check(self._lib.wass_triangulate(
            self._h, disp_roi.ctypes.data, W, H, (C.c_int * 4)(*roi_l), (C.c_int * 4)(*roi_r), C.byref(geom),
            right_img.ctypes.data, iw, ih, lm.ctypes.data if lm is not None else None,
            rm.ctypes.data if rm is not None else None, C.byref(tp), C.byref(h), C.byref(n)))
        return Mesh(self, h), int(n.value)

    def triangulate_dev(self, d_disp_roi, W, H, roi_l, roi_r, geom: Geom, d_right_img, d_left_mask=None,
                        d_right_mask=None, min_angle_deg=20.0, bbox=None, cam_distance=1.0, count=True):
        """count=False skips the point count (and with it the only host synchronisation of this call); the
        returned count is then None."""
        ih, iw = d_right_img.shape
        tp = TriParams(min_angle_deg, (C.c_double * 4)(*(bbox or (0, 0, iw, ih))), cam_distance)
        h = C.c_void_p(); n = C.c_uint64()
        self._check(self._lib.wass_triangulate_dev(
            self._h, d_disp_roi.data_ptr(), W, H, (C.c_int * 4)(*roi_l), (C.c_int * 4)(*roi_r), C.byref(geom),
            d_right_img.data_ptr(), iw, ih, d_left_mask.data_ptr() if d_left_mask is not None else None,
            d_right_mask.data_ptr() if d_right_mask is not None else None, C.byref(tp), C.byref(h),
            C.byref(n) if count else None))
        return Mesh(self, h), (int(n.value) if count else None)

    def mesh_upload(self, valid, p3d, gray=None):
        valid = np.ascontiguousarray(valid, np.uint8)
        p3d = np.ascontiguousarray(p3d, np.float64)
        hh, ww = valid.shape
        g = np.ascontiguousarray(gray, np.uint8) if gray is not None else None
        h = C.c_void_p()
        self._check(self._lib.wass_mesh_upload(self._h, ww, hh, valid.ctypes.data, p3d.ctypes.data,
                                               g.ctypes.data if g is not None else None, C.byref(h)))
        return Mesh(self, h)


def make_geom(g: dict, use_custom=False, disparity_compensation=0.0, dense_scale=1.0) -> Geom:
    """Fill a wass_geom from a dict of numpy matrices (keys as in synth.rig_geometry)."""
    G = Geom()
    for k in ("K_left", "K_right", "R", "T", "R1", "R2", "P1", "P2", "HLi", "HRi"):
        getattr(G, k)[:] = np.asarray(g[k], np.float64).ravel().tolist()
    G.use_custom = int(use_custom)
    G.disparity_compensation = disparity_compensation
    G.dense_scale = dense_scale
    return G


def ransac_sample(width: int, height: int, rounds: int, seed: int) -> np.ndarray:
    """srand(seed) + the reference's sampling loop (PovMesh.cpp:680-691), drawn from the library's private restatement of
    glibc's generator (libc rand() itself is shared with every thread of the process, the HIP runtime's included)."""
    lib = _lib.load()
    uv = np.empty((rounds, 6), np.int32)
    rc = lib.wass_ransac_sample_seeded(int(seed) & 0xFFFFFFFF, width, height, rounds, uv.ctypes.data)
    if rc != 0:
        raise WassError(rc, "wass_ransac_sample_seeded failed")
    return uv


def RT_from_plane(plane):
    lib = _lib.load()
    R = (C.c_double * 9)(); T = (C.c_double * 3)(); Ri = (C.c_double * 9)(); Ti = (C.c_double * 3)()
    lib.wass_RT_from_plane((C.c_double * 4)(*plane), R, T, Ri, Ti)
    return np.array(R[:]).reshape(3, 3), np.array(T[:]), np.array(Ri[:]).reshape(3, 3), np.array(Ti[:])


def planes_mean_accumulate(planes, acc=None) -> np.ndarray:
    lib = _lib.load()
    planes = np.ascontiguousarray(planes, np.float64).reshape(-1, 4)
    a = (C.c_double * 5)(*(acc if acc is not None else [0.0] * 5))
    lib.wass_planes_mean_accumulate(planes.ctypes.data_as(C.POINTER(C.c_double)), len(planes), a)
    return np.array(a[:])


def planes_mean_finish(acc5):
    lib = _lib.load()
    out = (C.c_double * 4)(); n = C.c_int()
    lib.wass_planes_mean_finish((C.c_double * 5)(*acc5), out, C.byref(n))
    return np.array(out[:]), int(n.value)


def stereo_rectify(K_left, K_right, width: int, height: int, R, T, alpha: float = 1.0) -> dict:
    """cv::stereoRectify(K_left, 0, K_right, 0, size, R, T, ..., flags=0, alpha, size) (wass_stereo.cpp:541)."""
    lib = _lib.load()
    d = lambda a: (C.c_double * len(np.ravel(a)))(*np.asarray(a, float).ravel())  # noqa: E731
    R1 = (C.c_double * 9)(); R2 = (C.c_double * 9)(); P1 = (C.c_double * 12)(); P2 = (C.c_double * 12)()
    r1 = (C.c_int * 4)(); r2 = (C.c_int * 4)()
    rc = lib.wass_stereo_rectify(d(K_left), d(K_right), width, height, d(R), d(T), alpha, R1, R2, P1, P2, r1, r2)
    if rc != _lib.WASS_OK:
        raise WassError(rc, "stereo_rectify: invalid rig (zero baseline?)")
    return dict(R1=np.array(R1[:]).reshape(3, 3), R2=np.array(R2[:]).reshape(3, 3), P1=np.array(P1[:]).reshape(3, 4),
                P2=np.array(P2[:]).reshape(3, 4), roi1=tuple(r1[:]), roi2=tuple(r2[:]))


def init_rectify_map(K, R, P, width: int, height: int):
    """cv::initUndistortRectifyMap(K, 0, R, P, size, CV_32FC1) (wass_stereo.cpp:600-601) -> (map_x, map_y)."""
    lib = _lib.load()
    d = lambda a: (C.c_double * len(np.ravel(a)))(*np.asarray(a, float).ravel())  # noqa: E731
    mx = np.empty((height, width), np.float32); my = np.empty((height, width), np.float32)
    rc = lib.wass_init_rectify_map(d(K), d(R), d(P), width, height, mx.ctypes.data, my.ctypes.data)
    if rc != _lib.WASS_OK:
        raise WassError(rc, "init_rectify_map: singular P*R")
    return mx, my


class Mesh:
    """Device-resident organised point cloud (wass_mesh) -- the PovMesh of the reference."""

    def __init__(self, ctx: Context, handle):
        self.ctx, self._h = ctx, handle
        w = C.c_int(); h = C.c_int()
        ctx._lib.wass_mesh_size(handle, C.byref(w), C.byref(h))
        self.width, self.height = w.value, h.value

    def close(self):
        if self._h:
            self.ctx._lib.wass_mesh_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def download(self):
        n = self.width * self.height
        valid = np.empty((self.height, self.width), np.uint8)
        p3d = np.empty((self.height, self.width, 3), np.float64)
        gray = np.empty((self.height, self.width), np.uint8)
        self.ctx._check(self.ctx._lib.wass_mesh_download(self.ctx._h, self._h, valid.ctypes.data, p3d.ctypes.data,
                                                         gray.ctypes.data))
        return valid, p3d, gray

    def grid_idw(self, plane, baseline: float, xmin: float, xmax: float, ymin: float, ymax: float, width: int, height: int, cell: str = "mean"):
        """Surface grid of this cloud aligned on `plane` (wassgridsurface.py:316-365, IDW): (float32 grid, uint8 mask).
        cell: "mean" or "median" -- the statistic a cell takes of its points."""
        gs = _lib.GridSetup()
        R, T, _, _ = RT_from_plane(plane)
        gs.R[:] = np.asarray(R, float).ravel().tolist(); gs.T[:] = np.asarray(T, float).ravel().tolist()
        gs.baseline, gs.xmin, gs.xmax, gs.ymin, gs.ymax, gs.width, gs.height = baseline, xmin, xmax, ymin, ymax, width, height
        grid = np.empty((height, width), np.float32); mask = np.empty((height, width), np.uint8)
        self.ctx._check(self.ctx._lib.wass_mesh_grid_idw_ex(self.ctx._h, self._h, C.byref(gs), {"mean": 0, "median": 1}[cell], grid.ctypes.data,
                                                            mask.ctypes.data))
        return grid, mask

    def zgap_percentile(self, pct: float):
        out = C.c_double(); n = C.c_uint64()
        self.ctx._check(self.ctx._lib.wass_mesh_zgap_percentile(self.ctx._h, self._h, pct, C.byref(out), C.byref(n)))
        return out.value, int(n.value)

    def keep_biggest_component(self, zgap: float) -> int:
        n = C.c_uint64()
        self.ctx._check(self.ctx._lib.wass_mesh_keep_biggest_component(self.ctx._h, self._h, zgap, C.byref(n)))
        return int(n.value)

    def remove_outliers(self, pct: float):
        """zgap_percentile + keep_biggest_component in one call (device-side decisions) -> (zgap, n_gaps, size)."""
        z = C.c_double(); n = C.c_uint64(); sz = C.c_uint64()
        self.ctx._check(self.ctx._lib.wass_mesh_remove_outliers(self.ctx._h, self._h, pct, C.byref(z), C.byref(n), C.byref(sz)))
        return z.value, int(n.value), int(sz.value)

    def fit_plane(self, uv, ransac_thr=1.0, max_distance=1.5, xmin=-9999., xmax=9999., ymin=-9999., ymax=9999.,
                  refine_max_distance=70.0, weight_by_distance=True, central_third_only=False) -> PlaneResult:
        """ransac_plane -> crop_plane -> refine_plane -> crop_plane in one call (wass_stereo.cpp:2062-2107)."""
        uv = np.ascontiguousarray(uv, np.int32)
        rp = RefineParams(xmin, xmax, ymin, ymax, refine_max_distance, int(weight_by_distance), int(central_third_only))
        res = PlaneResult()
        self.ctx._check(self.ctx._lib.wass_mesh_fit_plane(self.ctx._h, self._h, uv.ctypes.data, len(uv), ransac_thr,
                                                          C.byref(rp), max_distance, C.byref(res)))
        return res

    def finish_frame_async(self, uv, dst_ptr: int, capacity: int, percentile=99.0, ransac_thr=1.0, max_distance=1.5,
                           xmin=-9999., xmax=9999., ymin=-9999., ymax=9999., refine_max_distance=70.0, weight_by_distance=True,
                           central_third_only=False, inliers_ptr: int = 0, inliers_capacity: int = 0, inliers_every: int = 10,
                           inliers_text_ptr: int = 0, inliers_text_capacity: int = 0, component_mask_ptr: int = 0) -> None:
        """remove_outliers -> fit_plane -> xyzC encode + download, enqueued without a host synchronisation
        (wass_stereo.cpp:2046-2123); Context.frame_result() waits and reports.  inliers_ptr (pinned, capacity in points of 3 doubles):
        also every inliers_every-th refinement inlier; inliers_text_ptr (pinned, 40 bytes per point): also the TEXT of
        plane_refinement_inliers.xyz, formatted on the device (wass_mesh_finish_frame_async_ex2)."""
        uv = np.ascontiguousarray(uv, np.int32)
        rp = RefineParams(xmin, xmax, ymin, ymax, refine_max_distance, int(weight_by_distance), int(central_third_only))
        if inliers_ptr or inliers_text_ptr or component_mask_ptr:
            # component_mask_ptr (pinned, one byte per grid point): what the outlier removal left valid -- also kept on the device for
            # the debug pictures (debug_pictures_async)
            self.ctx._check(self.ctx._lib.wass_mesh_finish_frame_async_ex2(self.ctx._h, self._h, percentile, uv.ctypes.data, len(uv), ransac_thr, C.byref(rp),
                                                                           max_distance, dst_ptr, capacity, inliers_ptr or None, inliers_capacity, inliers_every,
                                                                           component_mask_ptr or None,
                                                                           inliers_text_ptr or None, inliers_text_capacity))
            return
        self.ctx._check(self.ctx._lib.wass_mesh_finish_frame_async(self.ctx._h, self._h, percentile, uv.ctypes.data, len(uv),
                                                                   ransac_thr, C.byref(rp), max_distance, dst_ptr, capacity))

    def debug_pictures_async(self, desc: "_lib.DebugDesc", dst_ptr: int, capacities) -> int:
        """The reference's eight debug pictures of this frame, rendered and JPEG-coded on the device behind the frame's tail
        (wass_debug_pictures_async; call after finish_frame_async(..., component_mask_ptr=...), before close()).  Returns the ticket
        for Context.debug_pictures_result."""
        caps = (C.c_size_t * 8)(*[int(c) for c in capacities])
        ticket = C.c_uint64()
        self.ctx._check(self.ctx._lib.wass_debug_pictures_async(self.ctx._h, self._h, C.byref(desc), dst_ptr, C.byref(caps), C.byref(ticket)))
        return int(ticket.value)

    def ransac_plane(self, uv, thr: float):
        uv = np.ascontiguousarray(uv, np.int32)
        plane = (C.c_double * 4)(); best = C.c_uint64(); found = C.c_int()
        self.ctx._check(self.ctx._lib.wass_mesh_ransac_plane(self.ctx._h, self._h, uv.ctypes.data, len(uv), thr, plane,
                                                             C.byref(best), C.byref(found)))
        return bool(found.value), np.array(plane[:]), int(best.value)

    def crop_plane(self, plane, thr: float) -> int:
        k = C.c_uint64()
        self.ctx._check(self.ctx._lib.wass_mesh_crop_plane(self.ctx._h, self._h, (C.c_double * 4)(*plane), thr, C.byref(k)))
        return int(k.value)

    def refine_plane(self, xmin=-9999., xmax=9999., ymin=-9999., ymax=9999., max_distance=70.0,
                     weight_by_distance=True, central_third_only=False):
        rp = RefineParams(xmin, xmax, ymin, ymax, max_distance, int(weight_by_distance), int(central_third_only))
        plane = (C.c_double * 4)(); n = C.c_uint64()
        self.ctx._check(self.ctx._lib.wass_mesh_refine_plane(self.ctx._h, self._h, C.byref(rp), plane, C.byref(n)))
        return np.array(plane[:]), int(n.value)

    def reject_codes(self):
        """(R0 codes, R1 codes) per grid pixel: why triangulate kept or rejected it (WASS_CODE_*; wass_stereo.cpp:1216-1338)."""
        out = np.empty((self.height, self.width), np.uint8)
        self.ctx._check(self.ctx._lib.wass_mesh_reject_codes(self.ctx._h, self._h, out.ctypes.data))
        return out & 15, out >> 4

    def refinement_inliers(self, every=10, xmin=-9999., xmax=9999., ymin=-9999., ymax=9999., max_distance=70.0,
                           central_third_only=False) -> np.ndarray:
        """The points of plane_refinement_inliers.xyz (wass_stereo.cpp:2077-2085): every `every`-th refinement inlier in raster
        order, selected on the device; (n, 3) float64."""
        rp = RefineParams(xmin, xmax, ymin, ymax, max_distance, 1, int(central_third_only))
        ptr = C.POINTER(C.c_double)(); n = C.c_uint64()
        self.ctx._check(self.ctx._lib.wass_mesh_refinement_inliers(self.ctx._h, self._h, C.byref(rp), int(every), C.byref(ptr), C.byref(n)))
        if not n.value:
            return np.zeros((0, 3))
        out = np.ctypeslib.as_array(ptr, shape=(n.value, 3)).copy()
        self.ctx._lib.wass_free(ptr)
        return out

    def encode_xyzc_to(self, plane, dst_ptr: int, capacity: int) -> int:
        """mesh_cam.xyzC bytes into a caller-owned host buffer (e.g. a pinned torch tensor); returns the size."""
        nb = C.c_size_t()
        pl = (C.c_double * 4)(*plane) if plane is not None else None
        self.ctx._check(self.ctx._lib.wass_mesh_encode_xyzc_to(self.ctx._h, self._h, pl, dst_ptr, capacity, C.byref(nb)))
        return int(nb.value)

    def encode_xyzc_async(self, plane, dst_ptr: int, capacity: int) -> int:
        """Like encode_xyzc_to, but the payload is still being copied when this returns; complete after
        Context.synchronize()."""
        nb = C.c_size_t()
        pl = (C.c_double * 4)(*plane) if plane is not None else None
        self.ctx._check(self.ctx._lib.wass_mesh_encode_xyzc_async(self.ctx._h, self._h, pl, dst_ptr, capacity, C.byref(nb)))
        return int(nb.value)

    def encode_xyzc(self, plane=None) -> bytes:
        buf = C.c_void_p(); nb = C.c_size_t()
        pl = (C.c_double * 4)(*plane) if plane is not None else None
        self.ctx._check(self.ctx._lib.wass_mesh_encode_xyzc(self.ctx._h, self._h, pl, C.byref(buf), C.byref(nb)))
        try:
            return C.string_at(buf, nb.value)
        finally:
            self.ctx._lib.wass_free(buf)
