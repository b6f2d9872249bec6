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
