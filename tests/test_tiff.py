"""The TIFF and JPEG readers of wass_prepare's inputs (wass_amd/host/tiff.hpp; wasscli lists tif / tiff among the supported formats,
cli/wasscli/wasscli.py:47): files written by Pillow / libtiff in the variants cameras and converters produce must decode to
the grey picture cv::imread(IMREAD_GRAYSCALE) would give."""
import os
import subprocess

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
Image = pytest.importorskip("PIL.Image")


@pytest.fixture(scope="module")
def tool():
    src = os.path.join(HERE, "native", "tiff_check.cpp")
    deps = [src] + [os.path.join(HERE, "..", "wass_amd", "host", f) for f in ("tiff.hpp", "hostio.hpp", "jpeg_read.hpp")]
    out_dir = os.path.join(HERE, "native", "_build")
    os.makedirs(out_dir, exist_ok=True)
    exe = os.path.join(out_dir, "tiff_check")
    if not os.path.exists(exe) or os.path.getmtime(exe) < max(os.path.getmtime(d) for d in deps):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-Wall", "-Wno-unused-function", src, "-o", exe, "-lz"])
    return exe


def _decode(tool, path, tmp_path):
    out = tmp_path / "out.raw"
    r = subprocess.run([tool, str(path), str(out)], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(r.stderr.strip())
    blob = out.read_bytes()
    nl = blob.index(b"\n")
    w, h = map(int, blob[:nl].split())
    return np.frombuffer(blob[nl + 1:], np.uint8).reshape(h, w)


def _grey(w, h, seed):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    img = 120 + 80 * np.sin(xx / 9.0) * np.cos(yy / 13.0) + rng.normal(0, 6, (h, w))
    img[h // 3:h // 3 + 10, : w // 2] = 200                               # long runs: PackBits / LZW repeats
    return np.clip(img, 0, 255).astype(np.uint8)


@pytest.mark.parametrize("compression", [None, "tiff_lzw", "tiff_adobe_deflate", "packbits"])
@pytest.mark.parametrize("w,h", [(64, 48), (333, 257), (1200, 901)])
def test_grey_tiff_variants(tool, tmp_path, compression, w, h):
    img = _grey(w, h, w)
    path = tmp_path / "g.tif"
    Image.fromarray(img).save(path, compression=compression) if compression else Image.fromarray(img).save(path)
    np.testing.assert_array_equal(_decode(tool, path, tmp_path), img)


def test_lzw_with_noise_fills_the_code_table(tool, tmp_path):
    """Incompressible data drives the LZW table through every code width and several ClearCodes."""
    img = np.random.default_rng(0).integers(0, 256, (400, 600), dtype=np.uint8)
    path = tmp_path / "n.tif"
    Image.fromarray(img).save(path, compression="tiff_lzw")
    np.testing.assert_array_equal(_decode(tool, path, tmp_path), img)


def test_rgb_and_16_bit(tool, tmp_path):
    rgb = np.stack([_grey(120, 90, 1), _grey(120, 90, 2), _grey(120, 90, 3)], -1)
    path = tmp_path / "c.tif"
    Image.fromarray(rgb).save(path, compression="tiff_lzw")
    want = (rgb[..., 0].astype(int) * 4899 + rgb[..., 1].astype(int) * 9617 + rgb[..., 2].astype(int) * 1868 + 8192) >> 14   # BT.601, as for PNG
    np.testing.assert_array_equal(_decode(tool, path, tmp_path), want.astype(np.uint8))
    g16 = (_grey(100, 70, 5).astype(np.uint16) << 8) | 0x5A
    path16 = tmp_path / "d.tif"
    Image.fromarray(g16).save(path16)
    np.testing.assert_array_equal(_decode(tool, path16, tmp_path), (g16 >> 8).astype(np.uint8))


def test_png_still_goes_through_the_same_entry_point(tool, tmp_path):
    img = _grey(77, 55, 9)
    path = tmp_path / "p.png"
    Image.fromarray(img).save(path)
    np.testing.assert_array_equal(_decode(tool, path, tmp_path), img)


@pytest.mark.parametrize("mode,subsampling,w,h", [("L", None, 64, 48), ("L", None, 333, 257), ("RGB", 0, 100, 75), ("RGB", 1, 211, 97), ("RGB", 2, 640, 480),
                                                      ("RGB", 2, 17, 9)])
def test_baseline_jpeg_decodes_to_what_libjpeg_gives(tool, tmp_path, mode, subsampling, w, h):
    """cv::imread(IMREAD_GRAYSCALE) lets libjpeg decode to grey = the luminance plane alone; Pillow's draft mode "L" asks libjpeg
    for the same thing."""
    g = _grey(w, h, w + h)
    if mode == "L":
        src = Image.fromarray(g)
    else:
        src = Image.fromarray(np.stack([g, np.roll(g, 5, 1), 255 - g], -1))
    path = tmp_path / "x.jpg"
    kw = {} if subsampling is None else {"subsampling": subsampling}
    src.save(path, quality=92, **kw)
    im = Image.open(path)
    im.draft("L", (w, h))                                                   # libjpeg decodes straight to grey (JCS_GRAYSCALE), like cv::imread
    want = np.asarray(im)
    assert want.ndim == 2
    got = _decode(tool, path, tmp_path)
    assert got.shape == want.shape
    d = np.abs(got.astype(int) - want.astype(int))
    assert d.max() <= 2 and d.mean() < 0.3, (d.max(), d.mean())            # double-precision IDCT against libjpeg's integer IDCT


def test_jpeg_restart_intervals_and_progressive(tool, tmp_path):
    g = _grey(200, 120, 4)
    path = tmp_path / "r.jpg"
    try:
        Image.fromarray(g).save(path, quality=90, restart_marker_blocks=7)
        blob = path.read_bytes()
        if b"\xff\xdd" in blob:                                             # Pillow wrote a DRI segment
            d = np.abs(_decode(tool, path, tmp_path).astype(int) - np.asarray(Image.open(path)).astype(int))
            assert d.max() <= 2
    except TypeError:
        pass                                                                # this Pillow cannot write restart markers
    prog = tmp_path / "p.jpg"
    Image.fromarray(g).save(prog, quality=90, progressive=True)
    with pytest.raises(RuntimeError, match="progressive"):
        _decode(tool, prog, tmp_path)


def test_unsupported_files_are_refused_with_a_message(tool, tmp_path):
    bad = tmp_path / "t.tif"
    bad.write_bytes(b"II*\x00\x08\x00\x00\x00\x00\x00")
    with pytest.raises(RuntimeError):
        _decode(tool, bad, tmp_path)


@pytest.fixture(scope="module")
def tool_asan():
    """The same driver under AddressSanitizer + UBSan: a malformed file must end in an error message, not in an out-of-bounds read."""
    src = os.path.join(HERE, "native", "tiff_check.cpp")
    deps = [src] + [os.path.join(HERE, "..", "wass_amd", "host", f) for f in ("tiff.hpp", "hostio.hpp", "jpeg_read.hpp")]
    out_dir = os.path.join(HERE, "native", "_build")
    os.makedirs(out_dir, exist_ok=True)
    exe = os.path.join(out_dir, "tiff_check_asan")
    if not os.path.exists(exe) or os.path.getmtime(exe) < max(os.path.getmtime(d) for d in deps):
        subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                               "-Wno-unused-function", src, "-o", exe, "-lz"])
    return exe


def _segments(blob):
    """(marker, offset of the marker's 0xFF, segment length incl. the length field) of the header segments of a JPEG file"""
    out, p = [], 2
    while p + 4 <= len(blob):
        assert blob[p] == 0xFF
        m = blob[p + 1]
        ln = (blob[p + 2] << 8) | blob[p + 3]
        out.append((m, p, ln))
        if m == 0xDA:
            break
        p += 2 + ln
    return out


def test_malformed_jpeg_headers_are_refused_not_read_out_of_bounds(tool_asan, tmp_path):
    """Bad table selectors in the scan header (Td / Ta up to 15 index four-entry arrays), segments that end in the middle of
    a table, frame headers shorter than their component list, plus random corruption of the header bytes."""
    img = _grey(96, 64, 3)
    good = tmp_path / "good.jpg"
    Image.fromarray(np.dstack([img, img // 2, 255 - img])).save(good, quality=90)
    blob = bytearray(good.read_bytes())
    seg = {m: (p, ln) for m, p, ln in _segments(blob)}
    cases = {}
    sos, _ = seg[0xDA]
    b = bytearray(blob); b[sos + 6] = 0xF0; cases["td15"] = b                  # first component: Td = 15
    b = bytearray(blob); b[sos + 6] = 0x0F; cases["ta15"] = b                  # Ta = 15
    dht, ln = seg[0xC4]
    b = bytearray(blob); b[dht + 2:dht + 4] = (10).to_bytes(2, "big"); cases["dht_cut"] = b      # counts run past the segment
    dqt, ln = seg[0xDB]
    b = bytearray(blob); b[dqt + 4] |= 0x10; cases["dqt16_cut"] = b            # 16-bit entries in a segment sized for 8-bit ones
    sof, ln = seg[0xC0]
    b = bytearray(blob); b[sof + 2:sof + 4] = (9).to_bytes(2, "big"); cases["sof_short"] = b
    b = bytearray(blob); b[sof + 2:sof + 4] = (1).to_bytes(2, "big"); cases["len1"] = b
    cases["cut_in_header"] = bytearray(blob[:sos + 3])
    rng = np.random.default_rng(5)
    for k in range(150):
        b = bytearray(blob)
        for _ in range(int(rng.integers(1, 4))):
            b[int(rng.integers(2, sos + 14))] = int(rng.integers(0, 256))
        cases["rnd%d" % k] = b
    env = dict(os.environ, ASAN_OPTIONS="exitcode=99:detect_leaks=0", UBSAN_OPTIONS="halt_on_error=1:exitcode=98")
    for name, data in cases.items():
        path = tmp_path / (name + ".jpg")
        path.write_bytes(bytes(data))
        r = subprocess.run([tool_asan, str(path), str(tmp_path / "o.raw")], capture_output=True, text=True, env=env, timeout=60)
        assert r.returncode in (0, 1), (name, r.returncode, r.stderr[-1500:])
        if name in ("td15", "ta15", "dht_cut", "dqt16_cut", "sof_short", "len1", "cut_in_header"):
            assert r.returncode == 1 and r.stderr.strip(), name
