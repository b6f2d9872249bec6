"""The device-side JPEG encoder of the debug pictures (wass_amd/csrc/jpeg.hip) against the host writer (wass_amd/host/jpeg.hpp): both follow
csrc/jpeg_spec.h -- integer colour conversion, integer DCT, defined rounding, a restart marker after every row of blocks -- and must give
the same FILE, byte for byte; an independent decoder (Pillow / libjpeg) must show the picture that went in."""
import io
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def host_tool():
    src = os.path.join(HERE, "native", "jpeg_check.cpp")
    out_dir = os.path.join(HERE, "native", "_build")
    os.makedirs(out_dir, exist_ok=True)
    exe = os.path.join(out_dir, "jpeg_check_gpu_ref")
    subprocess.check_call(["g++", "-O2", "-std=c++17", src, "-o", exe])
    return exe


def _host_bytes(tool, img, q, tmp_path):
    h, w = img.shape[:2]
    ch = 1 if img.ndim == 2 else 3
    raw = tmp_path / "in.raw"
    raw.write_bytes(np.ascontiguousarray(img).tobytes())
    out = tmp_path / "host.jpg"
    subprocess.check_call([tool, str(raw), str(w), str(h), str(ch), str(out), str(q)])
    return out.read_bytes()


def _picture(w, h, ch, seed, noise=4.0):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    base = 120 + 70 * np.sin(xx / 17.0) * np.cos(yy / 11.0) + rng.normal(0, noise, (h, w))
    if ch == 1:
        return np.clip(base, 0, 255).astype(np.uint8)
    img = np.stack([base, 255 - base * 0.7, 60 + 0.5 * base + 40 * np.sin(yy / 29.0)], -1)
    img[h // 4:h // 4 + 9, w // 5:w // 5 + 40] = (255, 0, 0)
    return np.clip(img, 0, 255).astype(np.uint8)


@pytest.mark.parametrize("w,h,ch,q", [(1, 1, 1, 95), (8, 8, 3, 95), (17, 9, 3, 60), (64, 48, 1, 95), (333, 257, 1, 95), (100, 37, 3, 95), (640, 480, 3, 95),
                                     (1031, 517, 1, 100), (2456, 2058, 1, 95), (1228, 1029, 3, 95)])
def test_device_and_host_write_the_same_file(gpu_ctx, host_tool, tmp_path, w, h, ch, q):
    import torch
    from PIL import Image
    img = _picture(w, h, ch, seed=w + 7 * ch)
    got = gpu_ctx.jpeg_encode(torch.from_numpy(img).cuda(), q)
    want = _host_bytes(host_tool, img, q, tmp_path)
    assert len(got) == len(want) and got == want
    dec = np.asarray(Image.open(io.BytesIO(got))).astype(np.float64)
    assert dec.shape == img.shape
    mse = float(((dec - img) ** 2).mean())
    assert mse == 0 or 10 * np.log10(255.0 ** 2 / mse) > (36.0 if q >= 95 else 22.0)


def test_noise_and_extremes(gpu_ctx, host_tool, tmp_path):
    """white noise (long codes, many 0xFF bytes to stuff, the largest files), checkerboards (DC differences of 11 bits, AC amplitudes of 10),
    flat pictures (runs of end-of-block only), twice in a row (the scratch is reused)"""
    import torch
    rng = np.random.default_rng(3)
    yy, xx = np.mgrid[0:200, 0:264]
    cases = [rng.integers(0, 256, (200, 264), dtype=np.uint8), rng.integers(0, 256, (123, 77, 3), dtype=np.uint8),
             (((xx + yy) % 2) * 255).astype(np.uint8), np.where(xx % 16 < 8, 0, 255).astype(np.uint8), np.full((64, 64), 255, np.uint8),
             np.zeros((40, 56, 3), np.uint8), np.full((33, 19, 3), (255, 0, 0), np.uint8)]
    for rep in range(2):
        for k, img in enumerate(cases):
            for q in (95, 100, 30):
                got = gpu_ctx.jpeg_encode(torch.from_numpy(np.ascontiguousarray(img)).cuda(), q)
                assert got == _host_bytes(host_tool, img, q, tmp_path), (rep, k, q)
    assert b"\xff\x00" in gpu_ctx.jpeg_encode(torch.from_numpy(cases[0]).cuda(), 100)       # stuffing did happen


def test_bad_arguments_are_errors(gpu_ctx):
    import torch
    import wass_amd
    with pytest.raises(ValueError):
        gpu_ctx.jpeg_encode(torch.zeros(4, 4, dtype=torch.float32, device="cuda"))
    with pytest.raises(wass_amd.WassError):
        gpu_ctx.jpeg_encode(torch.zeros(4, 4, 2, dtype=torch.uint8, device="cuda"))


def _frame_on_device(ctx, w, h, D, frame_idx, roi_l, roi_r, W0, H0):
    """one small frame through the device chain (SGM on ROI-sized crops ... frame tail with the component mask); what the pictures are drawn from"""
    import torch
    import wass_amd
    from wass_amd import _lib, default_sgm_params, synth
    dev = torch.device("cuda", 0)
    p = default_sgm_params(D, ndirs=5)
    geom = wass_amd.make_geom(synth.rig_geometry(w, h))
    dr, dl = (torch.from_numpy(a).to(dev) for a in synth.make_pair(w, h, D, frame_idx=frame_idx))
    d16 = torch.empty((h, w), dtype=torch.int16, device=dev)
    dispf = torch.empty((h, w), dtype=torch.float32, device=dev)
    ctx.sgm_disparity_dev(dr, dl, p, d16)
    ctx.disparity_postprocess_dev(d16, p, 1, 2, 0, dispf)
    mesh, _ = ctx.triangulate_dev(dispf, W0, H0, roi_l, roi_r, geom, dr, None, None, 20.0, None, 1.0, count=False)
    desc = _lib.DebugDesc()
    desc.W0, desc.H0 = W0, H0
    desc.roi_l[:] = roi_l
    desc.roi_r[:] = roi_r
    desc.d_left_crop, desc.d_right_crop, desc.d_disp16, desc.d_dispf = dl.data_ptr(), dr.data_ptr(), d16.data_ptr(), dispf.data_ptr()
    desc.num_disp, desc.min_disp, desc.disp_offset, desc.disparity_compensation, desc.quality = p.num_disp, p.min_disp, p.disp_offset, 0.0, 95
    return dict(p=p, dr=dr, dl=dl, d16=d16, dispf=dispf, mesh=mesh, desc=desc)


def test_the_eight_pictures_through_the_c_abi(gpu_ctx):
    """wass_debug_pictures_async on a small frame whose ROIs sit inside a larger picture: sizes and decodability of all eight, three of
    them against an independent numpy rendering (stereo_input: the padded inputs; disparity_final_scaled: render_disparity_float;
    stereo: pasted crops, red rectangles, a red line every 20 rows), a slot that is too small, a destination that is not pinned, tickets."""
    import torch
    import wass_amd
    from PIL import Image
    w, h, D = 320, 240, 64
    W0, H0 = 400, 300
    roi_l, roi_r = (40, 30, w, h), (52, 30, w, h)
    uv = wass_amd.ransac_sample(w, h, 300, 7)
    gpu_ctx.set_tail_overlap(True)
    try:
        f = _frame_on_device(gpu_ctx, w, h, D, 11, roi_l, roi_r, W0, H0)
        pin = torch.zeros(148 + 6 * w * h, dtype=torch.uint8).pin_memory()
        cc = torch.zeros(w * h, dtype=torch.uint8).pin_memory()
        sizes = [(2 * W0, H0, 3), (w + D, 2 * h, 1), (w, h, 1), (w, h, 1), ((W0 + 1) // 2, (H0 + 1) // 2, 3), (W0, H0, 3), (W0, H0, 3), ((w + 1) // 2, (h + 1) // 2, 3)]
        from wass_amd import _lib
        import ctypes as C
        for k, want in enumerate(sizes):
            a, b, c = C.c_int(), C.c_int(), C.c_int()
            assert _lib.load().wass_debug_picture_size(C.byref(f["desc"]), k, C.byref(a), C.byref(b), C.byref(c)) == 0 and (a.value, b.value, c.value) == want
        caps = [((ww * hh * (3 if c == 3 else 2) // 2 + 65536 + 63) // 64) * 64 for ww, hh, c in sizes]
        caps[5] = 1024                                                # R0 cannot fit: reported, not written, the others unharmed
        dst = torch.zeros(sum(caps), dtype=torch.uint8).pin_memory()
        f["mesh"].finish_frame_async(uv, pin.data_ptr(), pin.numel(), component_mask_ptr=cc.data_ptr())
        with pytest.raises(wass_amd.WassError):
            f["mesh"].debug_pictures_async(f["desc"], torch.zeros(sum(caps), dtype=torch.uint8).data_ptr(), caps)      # pageable memory
        t1 = f["mesh"].debug_pictures_async(f["desc"], dst.data_ptr(), caps)
        f["mesh"].close()
        fr = gpu_ctx.frame_result()
        assert fr.found
        n = gpu_ctx.debug_pictures_result(t1)
        assert n[5] == 0 and all(v > 600 for i, v in enumerate(n) if i != 5)
        pics, off = [], 0
        for k, (ww, hh, c) in enumerate(sizes):
            if n[k]:
                blob = dst[off:off + n[k]].numpy().tobytes()
                assert blob[:2] == b"\xff\xd8" and blob[-2:] == b"\xff\xd9"
                im = np.asarray(Image.open(io.BytesIO(blob)))
                assert im.shape == ((hh, ww) if c == 1 else (hh, ww, 3)), k
                pics.append(im.astype(int))
            else:
                pics.append(None)
            off += caps[k]
        dr, dl, dispf = f["dr"].cpu().numpy(), f["dl"].cpu().numpy(), f["dispf"].cpu().numpy()
        # stereo_input.jpg: left above right, zero-padded to w + D (offset 0)
        want = np.zeros((2 * h, w + D), int)
        want[:h, D:D + w] = dl
        want[h:, D:D + w] = dr
        assert np.abs(pics[1] - want).mean() < 2.5
        # disparity_final_scaled.jpg
        mn, mx = min(np.float32(w + 1), dispf.min()), max(np.float32(0), dispf.max())
        want = ((dispf - mn) / (mx - mn) * np.float32(255.0)).astype(np.uint8).astype(int)
        assert np.abs(pics[3] - want).mean() < 2.5
        # stereo.jpg: red lines, red rectangles, the crops pasted at their ROIs, black elsewhere
        st = pics[0]
        assert (st[::20, :, 0] > 200).all() and (st[::20, :, 1] < 70).all()
        body = np.ones((H0, 2 * W0), bool)
        body[::20] = False
        left = np.zeros((H0, W0), int); left[roi_l[1]:roi_l[1] + h, roi_l[0]:roi_l[0] + w] = dl[:H0 - roi_l[1], :W0 - roi_l[0]]
        right = np.zeros((H0, W0), int); right[roi_r[1]:roi_r[1] + h, roi_r[0]:roi_r[0] + w] = dr[:H0 - roi_r[1], :W0 - roi_r[0]]
        grey = np.concatenate([left, right], 1)
        inner = np.zeros_like(body)
        for x0, roi in ((0, roi_l), (W0, roi_r)):
            inner[roi[1] + 3:roi[1] + min(h, H0 - roi[1]) - 3, x0 + roi[0] + 3:x0 + roi[0] + min(w, W0 - roi[0]) - 3] = True
        sel = body & inner
        assert np.abs(st[..., 1][sel] - grey[sel]).mean() < 3.0
        assert (st[roi_l[1], roi_l[0] + 5:roi_l[0] + 50, 0] > 200).all() and (st[roi_l[1], roi_l[0] + 5:roi_l[0] + 50, 2] < 80).all()     # the rectangle's top edge
        assert st[5, 5].max() < 12                                                                                                   # outside the ROI: black
        # graph_components.jpg: green where the component mask kept a point
        gc = pics[7]
        keep = cc.numpy().reshape(h, w)[::2, ::2][:gc.shape[0], :gc.shape[1]] > 0
        assert keep.mean() > 0.2 and (gc[..., 1][keep] > 128).mean() > 0.9
        # tickets: four are kept
        with pytest.raises(wass_amd.WassError):
            gpu_ctx.debug_pictures_result(t1 + 1)
        assert gpu_ctx.debug_pictures_result(t1) == n
    finally:
        gpu_ctx.set_tail_overlap(False)
