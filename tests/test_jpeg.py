"""The baseline JPEG writer of the debug pictures (wass_amd/host/jpeg.hpp): the files must be
decodable by an independent decoder (Pillow / libjpeg) and show the picture that went in."""
import os
import subprocess

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
Image = pytest.importorskip("PIL.Image")


@pytest.fixture(scope="module")
def tool():
    src = os.path.join(HERE, "native", "jpeg_check.cpp")
    hdr = os.path.join(HERE, "..", "wass_amd", "host", "jpeg.hpp")
    out_dir = os.path.join(HERE, "native", "_build")
    os.makedirs(out_dir, exist_ok=True)
    exe = os.path.join(out_dir, "jpeg_check")
    if not os.path.exists(exe) or os.path.getmtime(exe) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-Wall", src, "-o", exe])
    return exe


def _picture(w, h, ch, seed):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    base = 120 + 70 * np.sin(xx / 17.0) * np.cos(yy / 11.0) + rng.normal(0, 4, (h, w))
    if ch == 1:
        img = base
    else:
        img = np.stack([base, 255 - base * 0.7, 60 + 0.5 * base + 40 * np.sin(yy / 29.0)], -1)
        img[h // 4:h // 4 + 9, w // 5:w // 5 + 40] = (255, 0, 0)          # saturated flat patches, like the colour codes of R0 / R1
        img[h // 2:h // 2 + 12, w // 3:w // 3 + 25] = (0, 255, 255)
    return np.clip(img, 0, 255).astype(np.uint8)


@pytest.mark.parametrize("w,h,ch,q", [(64, 48, 1, 95), (333, 257, 1, 95), (100, 37, 3, 95), (640, 480, 3, 95), (17, 9, 3, 60), (8, 8, 1, 100), (1, 1, 3, 95)])
def test_jpeg_files_decode_to_the_picture(tool, tmp_path, w, h, ch, q):
    img = _picture(w, h, ch, seed=w + ch)
    raw = tmp_path / "in.raw"
    raw.write_bytes(img.tobytes())
    out = tmp_path / "out.jpg"
    subprocess.check_call([tool, str(raw), str(w), str(h), str(ch), str(out), str(q)])
    blob = out.read_bytes()
    assert blob[:4] == b"\xff\xd8\xff\xe0" and blob[6:11] == b"JFIF\0" and blob[-2:] == b"\xff\xd9"
    im = Image.open(out)
    im.load()
    assert im.size == (w, h) and im.mode == ("L" if ch == 1 else "RGB")
    got = np.asarray(im).astype(np.float64)
    err = got - img.astype(np.float64)
    mse = float((err ** 2).mean())
    psnr = 99.0 if mse == 0 else 10 * np.log10(255.0 ** 2 / mse)
    assert psnr > (38.0 if q >= 95 else 22.0), psnr            # quality 95 without chroma subsampling; 60 on a tiny noisy picture
    assert np.abs(err).max() <= (24 if q >= 95 else 80)


def test_jpeg_handles_extreme_blocks(tool, tmp_path):
    """Checkerboards and black / white steps produce the largest coefficients (DC differences of size 11, AC of size 10)."""
    w, h = 64, 64
    yy, xx = np.mgrid[0:h, 0:w]
    img = (((xx + yy) % 2) * 255).astype(np.uint8)
    img[:, 32:] = np.where(xx[:, 32:] % 16 < 8, 0, 255)
    raw = tmp_path / "in.raw"; raw.write_bytes(img.tobytes())
    out = tmp_path / "out.jpg"
    subprocess.check_call([tool, str(raw), str(w), str(h), "1", str(out), "100"])
    got = np.asarray(Image.open(out)).astype(int)
    assert np.abs(got - img.astype(int)).mean() < 6
