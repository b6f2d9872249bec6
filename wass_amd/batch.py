"""Frame-parallel driver: the MI355X replacement for wasscli's process fan-out.

The reference runs one `wass_stereo` process per frame, N at a time (cli/wasscli/wasscli.py:305-364), and joins the
per-frame `plane.txt` files into `output/planes.txt` (:341-343); the gridding stage later takes np.nanmean of that
file as the sequence's mean sea plane (gridding/wassgridsurface/wassgridsurface.py:672-678).

Here frames are sharded over ranks (one process per GPU, frame i -> rank i mod world); there is no data-path
collective.  The only exchange is Coll-1: an all-reduce of [sum a, sum b, sum c, sum d, n_valid] (5 doubles) and a
gather of the per-frame planes so that rank 0 can write planes.txt in frame order.  Backend "nccl" is RCCL on ROCm;
"gloo" is used by the CPU tests.
"""
from __future__ import annotations

from typing import Callable, Sequence

import numpy as np

from . import stereo


class FrameOutput:
    """What one frame leaves behind: the refined plane (NaNs when RANSAC or the refinement failed, as plane.txt
    has them), the bytes of mesh_cam.xyzC (a view of the pipeline's pinned buffer: valid until two more frames
    have been submitted) and the stage-by-stage numbers of wass_frame_result."""

    def __init__(self, index, result, xyzc):
        self.index, self.result, self.xyzc = index, result, xyzc
        ok = bool(result.found and result.refine_ok)
        self.plane = np.array(result.plane[:]) if ok else np.full(4, np.nan)
        self.n_points = int(result.n_points)
        # 1: the frame's block costs left the int16 range in which the reference is defined; the
        # synchronous entry point reports the same condition as WASS_ERR_COST_OVERFLOW
        self.cost_overflow = int(result.sgm_cost_overflow)
        self.sgm_timeout = int(result.sgm_timeout)


class FramePipeline:
    """Frames back to back on one context without host round-trips (the GPU counterpart of one wasscli worker).

    submit() enqueues a whole frame -- SGM on the context's main stream, disparity clean-up, triangulation, outlier
    removal, plane fit and the xyzC encoder on its tail stream, the file image on its copy stream -- and returns the
    output of the frame BEFORE LAST (two frames stay pending; None for the first two calls), whose download finished a
    whole frame ago.  drain() returns the outputs still pending, in order; flush() the last of them.  Lifetime contract: the tail of frame i keeps reading d_right_image and the masks while frame i+1 is
    being prepared, so the caller must leave every input of frame i untouched until submit() of frame i+2 has
    returned (or flush()); ordering against the stream that produced the inputs is handled here.  Inputs are rectified crops resident in HBM (torch uint8 CUDA tensors); stage parameters are the
    defaults of wass_stereo unless given.
    """

    def __init__(self, ctx: "stereo.Context", width: int, height: int, params, geom, roi_l=None, roi_r=None,
                 dilate_steps=1, erode_steps=2, median_wsize=0, min_angle_deg=20.0, zgap_percentile=99.0,
                 ransac_rounds=400, random_seed=12345, ransac_thr=1.0, plane_max_distance=1.5, refine=None,
                 tail_overlap=True, inliers_text=False, keep_inlier_points=False):
        import torch
        if params.dense_scale != 1.0:
            # the pipelined chain keeps every map at the crop size; DENSE_SCALE != 1 goes through the stage-by-stage calls
            # (Context.sgm_disparity + disparity_postprocess with the scale), as the drop-in wass_stereo does
            raise ValueError("FramePipeline supports DENSE_SCALE = 1 only")
        self.ctx, self.w, self.h, self.params, self.geom = ctx, width, height, params, geom
        self.roi_l = tuple(roi_l) if roi_l is not None else (0, 0, width, height)
        self.roi_r = tuple(roi_r) if roi_r is not None else (0, 0, width, height)
        self.dilate, self.erode, self.median, self.min_angle = dilate_steps, erode_steps, median_wsize, min_angle_deg
        self.pct, self.rounds, self.seed = zgap_percentile, ransac_rounds, random_seed
        self.ransac_thr, self.max_distance, self.refine = ransac_thr, plane_max_distance, dict(refine or {})
        dev = torch.device("cuda", ctx.device_id)
        # two disparity buffers, alternated: the clean-up of frame i reads one while the SGM stage of frame i+1 writes
        # the other; two pinned file images, alternated: frame i's is read by the caller while frame i+1's is written
        self._disp16 = [torch.empty((height, width), dtype=torch.int16, device=dev) for _ in range(2)]
        self.NBUF = 4                                      # output sets: two frames pending, one in the caller's hands, one being enqueued
        self._dispf = torch.empty((height, width), dtype=torch.float32, device=dev)
        mw, mh = self.roi_r[2], self.roi_r[3]
        self._host = [torch.empty(148 + 6 * mw * mh, dtype=torch.uint8, pin_memory=True) for _ in range(self.NBUF)]
        # wass_stereo seeds rand() once per process = once per frame (wass_stereo.cpp:1864-1872), so every frame of a
        # sequence draws the same RANSAC triplets for a given grid size: draw them once
        self._uv = stereo.ransac_sample(mw, mh, ransac_rounds, random_seed)
        # plane_refinement_inliers.xyz (wass_stereo.cpp:2077-2085) as the C++ driver produces it: every 10th refinement inlier, and
        # the file's text formatted on the device; off by default (the reference's debug artefact is not part of the metric's pass)
        self._inl_text = None
        self._keep_points = keep_inlier_points
        if inliers_text:
            cap = (mw * mh + 9) // 10
            self._inl_text = [torch.empty(cap * 40, dtype=torch.uint8, pin_memory=True) for _ in range(self.NBUF)]
            self._inl_cap = cap
        self._n = 0
        self._pending = []                                 # submitted, record not read yet: at most two (the library's limit)
        ctx.set_tail_overlap(tail_overlap)

    def submit(self, d_right, d_left, d_right_image=None, d_left_mask=None, d_right_mask=None):
        """d_right/d_left: rectified crops; d_right_image: the undistorted right image sampled for the point colour
        (defaults to d_right); masks: 0/1 uint8 images of the originals' size or None."""
        import torch
        ctx, k = self.ctx, self._n & 1
        out = self._disp16[k]
        # the inputs were produced on torch's current stream; the context runs on streams of its own
        ctx.wait_for_stream(torch.cuda.current_stream(ctx.device_id).cuda_stream)
        ctx.sgm_disparity_dev(d_right, d_left, self.params, out)
        ctx.disparity_postprocess_dev(out, self.params, self.dilate, self.erode, self.median, self._dispf)
        img = d_right_image if d_right_image is not None else d_right
        mesh, _ = ctx.triangulate_dev(self._dispf, self.w, self.h, self.roi_l, self.roi_r, self.geom, img, d_left_mask,
                                      d_right_mask, self.min_angle, None, 1.0, count=False)
        # two frames stay pending (the library's limit): the record read here belongs to a tail that ended a whole frame ago, so this
        # thread does not wait and the next frame's SGM stage is queued before the current one has finished
        prev = self._collect() if len(self._pending) >= 2 else None
        kb = self._n % self.NBUF
        host = self._host[kb]
        extra = {}
        if self._inl_text is not None:
            extra = dict(inliers_capacity=self._inl_cap, inliers_every=10,
                         inliers_text_ptr=self._inl_text[kb].data_ptr(), inliers_text_capacity=self._inl_text[kb].numel())
        mesh.finish_frame_async(self._uv, host.data_ptr(), host.numel(), self.pct, self.ransac_thr, self.max_distance,
                                **self.refine, **extra)
        mesh.close()
        self._pending.append((self._n, host, kb))
        self._n += 1
        return prev

    def _collect(self):
        if not self._pending:
            return None
        idx, host, kb = self._pending.pop(0)
        fr = self.ctx.frame_result()                       # the OLDEST pending frame's record
        out = FrameOutput(idx, fr, host[:int(fr.xyzc_bytes)].numpy())
        if self._inl_text is not None:
            # the bytes of plane_refinement_inliers.xyz (valid like xyzc: until two more frames have been submitted); None: a number
            # was outside the device formatter's domain and the caller has to format inliers_xyz itself.  The points stay on the
            # device unless they are needed for that (or asked for): fetched here, before the frame after next reuses the buffer
            out.inliers_text = None if fr.inliers_text_unsupported else self._inl_text[kb][:int(fr.inliers_text_bytes)].numpy()
            out.inliers_xyz = self.ctx.frame_inliers(fr.n_inliers_out) if (fr.inliers_text_unsupported or self._keep_points) else None
        return out

    def drain(self):
        """Wait for every submitted frame; their outputs in submission order."""
        outs = []
        while self._pending:
            outs.append(self._collect())
        return outs

    def flush(self):
        """Wait for every submitted frame and return the LAST one's output (drain() returns all of them)."""
        outs = self.drain()
        return outs[-1] if outs else None


def shard(n_frames: int, rank: int, world: int) -> list[int]:
    """Frame indices owned by `rank` (round-robin, like wasscli's work queue)."""
    return list(range(rank, n_frames, world))


def planes_text(planes: np.ndarray) -> str:
    """planes.txt: one frame per line, the four plane.txt values joined by single spaces (wasscli.py:341-343)."""
    lines = []
    for p in np.asarray(planes, float).reshape(-1, 4):
        lines.append("nan nan nan nan" if np.isnan(p).any() else " ".join(repr(float(v)) for v in p))
    return "\n".join(lines) + ("\n" if lines else "")


def run_sequence(n_frames: int, process_frame: Callable[[int], Sequence[float]], dist=None, device=None):
    """Process this rank's share of frames and reduce the planes.

    process_frame(i) -> plane (4 floats, NaNs when the plane fit failed; wass_stereo.cpp:2101-2107).
    Returns (mean_plane[4], n_valid, all_planes[n_frames][4] on rank 0 else None).
    """
    import torch
    rank = dist.get_rank() if dist is not None else 0
    world = dist.get_world_size() if dist is not None else 1
    mine = shard(n_frames, rank, world)
    local = np.full((len(mine), 4), np.nan)
    for k, i in enumerate(mine):
        local[k] = np.asarray(process_frame(i), float)
    acc = stereo.planes_mean_accumulate(local) if len(mine) else np.zeros(5)
    planes_all = None
    if dist is not None:
        t = torch.tensor(acc, dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)                       # Coll-1
        acc = t.cpu().numpy()
        # gather per-frame planes (padded to the largest shard) for planes.txt
        per = (n_frames + world - 1) // world
        buf = torch.full((per, 4), float("nan"), dtype=torch.float64, device=device)
        if len(mine):
            buf[:len(mine)] = torch.from_numpy(local).to(buf.device)
        out = [torch.empty_like(buf) for _ in range(world)]
        dist.all_gather(out, buf)
        if rank == 0:
            planes_all = np.full((n_frames, 4), np.nan)
            for r in range(world):
                idx = shard(n_frames, r, world)
                planes_all[idx] = out[r].cpu().numpy()[:len(idx)]
    else:
        planes_all = local
    mean, n_valid = stereo.planes_mean_finish(acc)
    return mean, n_valid, planes_all
