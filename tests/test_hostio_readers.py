"""The host PNG reader (wass_amd/host/hostio.hpp: read_png_gray, what cv::imread(IMREAD_GRAYSCALE) is to wass_stereo.cpp:393-396) on every
row filter of the PNG specification.  cv::imwrite -- i.e. the reference's wass_prepare -- writes Sub-filtered rows at zlib level 1; other
encoders choose a filter per row.  The fast path (None / Sub / Up straight into the picture) and the general path (Average, Paeth, colour)
may alternate row by row: the previous row has to be the right one whichever path produced it."""
import os
import struct
import subprocess
import zlib

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def dump(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("probe") / "host_probe")
    subprocess.check_call(["g++", "-O1", "-std=c++17", os.path.join(ROOT, "tests", "helpers", "host_probe.cpp"), "-o", exe, "-lz", "-ldl"])
    return exe


def _paeth(a, b, c):
    p = a + b - c
    pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
    return a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)


def _filter_rows(img, filters):
    """img (h, w, ch) uint8 -> the filtered scanlines of the PNG specification, filter type per row from `filters`"""
    h, w, ch = img.shape
    rows = img.reshape(h, w * ch).astype(np.int32)
    out = bytearray()
    for y in range(h):
        cur = rows[y]
        up = rows[y - 1] if y > 0 else np.zeros_like(cur)
        left = np.concatenate([np.zeros(ch, np.int32), cur[:-ch]])
        upleft = np.concatenate([np.zeros(ch, np.int32), up[:-ch]])
        ft = int(filters[y])
        if ft == 0:
            f = cur
        elif ft == 1:
            f = cur - left
        elif ft == 2:
            f = cur - up
        elif ft == 3:
            f = cur - (left + up) // 2
        else:
            f = cur - np.array([_paeth(int(a), int(b), int(c)) for a, b, c in zip(left, up, upleft)], np.int32)
        out.append(ft)
        out += (f & 255).astype(np.uint8).tobytes()
    return bytes(out)


def _write_png(path, img, filters, level=6, idat_split=1):
    h, w, ch = img.shape
    ctype = {1: 0, 2: 4, 3: 2, 4: 6}[ch]
    z = zlib.compress(_filter_rows(img, filters), level)

    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xFFFFFFFF)
    n = max(1, len(z) // idat_split)
    idat = b"".join(chunk(b"IDAT", z[i:i + n]) for i in range(0, len(z), n))
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, ctype, 0, 0, 0)) + chunk(b"tEXt", b"Comment\x00x") + idat + chunk(b"IEND", b""))


def _decode(dump, path):
    r = subprocess.run([dump, "png", path], capture_output=True)
    assert r.returncode == 0, r.stderr.decode()
    w, h = struct.unpack("ii", r.stdout[:8])
    return np.frombuffer(r.stdout[8:], np.uint8).reshape(h, w)


@pytest.mark.parametrize("which", ["none", "sub", "up", "average", "paeth", "mixed", "fast_then_general", "general_then_fast"])
def test_grey_rows_of_every_filter(dump, tmp_path, which):
    rng = np.random.default_rng(hash(which) % 1000)
    h, w = 37, 53
    img = (rng.integers(0, 256, (h, w, 1)) // 3 + np.arange(w)[None, :, None] * 2).astype(np.uint8)
    filters = {"none": [0] * h, "sub": [1] * h, "up": [2] * h, "average": [3] * h, "paeth": [4] * h, "mixed": rng.integers(0, 5, h),
               "fast_then_general": [1, 2, 0, 4, 3, 2, 1, 4] * 5, "general_then_fast": [4, 2, 3, 1, 4, 0, 3, 2] * 5}[which][:h]
    p = str(tmp_path / "g.png")
    _write_png(p, img, filters, idat_split=3)
    np.testing.assert_array_equal(_decode(dump, p), img[:, :, 0])


@pytest.mark.parametrize("ch", [2, 3, 4])
def test_colour_and_alpha_pictures_come_out_grey(dump, tmp_path, ch):
    """grey + alpha: the grey channel; RGB(A): cv::cvtColor's fixed-point luma (R 4899, G 9617, B 1868, 14 bits) as imread applies it"""
    rng = np.random.default_rng(ch)
    h, w = 23, 31
    img = rng.integers(0, 256, (h, w, ch), dtype=np.uint8)
    p = str(tmp_path / "c.png")
    _write_png(p, img, rng.integers(0, 5, h))
    got = _decode(dump, p)
    if ch == 2:
        want = img[:, :, 0]
    else:
        v = img.astype(np.int64)
        want = ((v[:, :, 0] * 4899 + v[:, :, 1] * 9617 + v[:, :, 2] * 1868 + 8192) >> 14).astype(np.uint8)
    np.testing.assert_array_equal(got, want)


def test_a_five_megapixel_sub_filtered_picture_like_cv_imwrite_writes(dump, tmp_path):
    rng = np.random.default_rng(9)
    h, w = 2058, 2456
    img = (rng.integers(0, 64, (h, w, 1)) + (np.arange(w)[None, :, None] % 190)).astype(np.uint8)
    rows = img.reshape(h, w).astype(np.int16)
    f = np.concatenate([np.ones((h, 1), np.int16), rows - np.concatenate([np.zeros((h, 1), np.int16), rows[:, :-1]], axis=1)], axis=1)
    z = zlib.compress((f & 255).astype(np.uint8).tobytes(), 1)

    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xFFFFFFFF)
    p = str(tmp_path / "big.png")
    with open(p, "wb") as fo:
        fo.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 0, 0, 0, 0)) + b"".join(chunk(b"IDAT", z[i:i + 8192]) for i in range(0, len(z), 8192)) + chunk(b"IEND", b""))
    for env in ({}, {"WASS_NO_LIBDEFLATE": "1"}):                        # libdeflate (dlopen) and the libz fall-back
        r = subprocess.run([dump, "png", p], capture_output=True, env=dict(os.environ, **env))
        assert r.returncode == 0, r.stderr.decode()
        np.testing.assert_array_equal(np.frombuffer(r.stdout[8:], np.uint8).reshape(h, w), img[:, :, 0])


def _py_decode(path):
    """an independent PNG decoder (struct + zlib: chunk CRCs and the stream's Adler-32 are checked), filter None only"""
    blob = open(path, "rb").read()
    assert blob[:8] == b"\x89PNG\r\n\x1a\n"
    pos, idat, w, h = 8, b"", 0, 0
    while pos < len(blob):
        n, t = struct.unpack(">I4s", blob[pos:pos + 8])
        d = blob[pos + 8:pos + 8 + n]
        assert zlib.crc32(t + d) & 0xFFFFFFFF == struct.unpack(">I", blob[pos + 8 + n:pos + 12 + n])[0]
        if t == b"IHDR":
            w, h, depth, ctype = struct.unpack(">IIBB", d[:10]); assert (depth, ctype) == (8, 0)
        elif t == b"IDAT":
            idat += d
        pos += 12 + n
    raw = np.frombuffer(zlib.decompress(idat), np.uint8).reshape(h, w + 1)
    assert (raw[:, 0] == 0).all()
    return raw[:, 1:], idat


@pytest.mark.parametrize("shape", [(1, 1), (3, 70000), (37, 53), (300, 437), (2058, 2456)])
def test_stored_pictures_as_the_products_wass_prepare_writes_them(dump, tmp_path, shape):
    """Round 6: undistorted/*.png are written as zlib streams of STORED blocks (write_png_gray level 0): valid PNG for every reader, and a
    frame's two inflates (2 x 17 ms of its 60 ms of host time) become copies.  The writer against an independent decoder; the reader's
    fast path (file bytes straight into the picture) on its own files, on zlib's level-0 streams cut into IDAT chunks anywhere, and on
    stored streams whose rows are filtered after all (general path)."""
    rng = np.random.default_rng(shape[0] * 7 + shape[1])
    h, w = shape
    img = rng.integers(0, 256, (h, w, 1), dtype=np.uint8)
    src = str(tmp_path / "src.png")
    _write_png(src, img, [1] * h, level=1)
    out0, out1 = str(tmp_path / "stored.png"), str(tmp_path / "l1.png")
    for out, level in ((out0, 0), (out1, 1)):
        assert subprocess.run([dump, "repng", src, out, str(level)]).returncode == 0
        px, idat = _py_decode(out)
        np.testing.assert_array_equal(px, img[:, :, 0])
        np.testing.assert_array_equal(_decode(dump, out), img[:, :, 0])
    raw_bytes = (w + 1) * h
    _, idat0 = _py_decode(out0)
    assert len(idat0) == raw_bytes + 5 * max(1, -(-raw_bytes // 65535)) + 6           # header, 5 bytes per stored block, Adler-32
    # zlib's own level-0 stream, IDAT cut into seven pieces
    p = str(tmp_path / "z0.png")
    _write_png(p, img, [0] * h, level=0, idat_split=7)
    np.testing.assert_array_equal(_decode(dump, p), img[:, :, 0])
    # stored blocks, filtered rows: not the fast path's business
    filt = rng.integers(0, 5, h) if h * w < 200000 else [1] * h
    _write_png(p, img, filt, level=0, idat_split=2)
    np.testing.assert_array_equal(_decode(dump, p), img[:, :, 0])


def test_broken_files_are_errors_not_crashes(dump, tmp_path):
    img = np.arange(20 * 30, dtype=np.uint8).reshape(20, 30, 1)
    p = str(tmp_path / "ok.png")
    _write_png(p, img, [1] * 20)
    blob = open(p, "rb").read()
    for name, bad in (("truncated", blob[:len(blob) // 2]), ("not_png", b"hello" * 100), ("empty", b""), ("bad_filter", None)):
        q = str(tmp_path / (name + ".png"))
        if bad is None:
            _write_png(q, img, [7] * 20)
        else:
            open(q, "wb").write(bad)
        r = subprocess.run([dump, "png", q], capture_output=True)
        assert r.returncode == 1 and r.stderr                            # an exception with a message, caught by the caller


def test_calibration_matrices_as_opencv_writes_them(dump, tmp_path):
    """cv::FileStorage's XML (what wass_prepare / the calibration tools leave in a workdir, wass_stereo.cpp:340-386): values like `1.`, `0.`,
    `2.4560000000000000e+03`, several per line with arbitrary indentation, `<dt>d</dt>` or `<dt>f</dt>`, a comment line in front."""
    text = """<?xml version="1.0"?>
<!-- written by cv::FileStorage -->
<opencv_storage>
<intr type_id="opencv-matrix">
  <rows>3</rows>
  <cols>3</cols>
  <dt>d</dt>
  <data>
    2.4560000000000000e+03 0. 1.2275000000000000e+03 0.
    2.4561234567890123e+03 1.0285000000000000e+03 0. 0. 1.</data></intr>
</opencv_storage>
"""
    p = tmp_path / "intrinsics_00000000.xml"
    p.write_text(text)
    r = subprocess.run([dump, "xml", str(p)], capture_output=True, text=True)
    assert r.returncode == 0
    lines = r.stdout.split()
    assert lines[:2] == ["3", "3"]
    assert [float(v) for v in lines[2:]] == [2456.0, 0.0, 1227.5, 0.0, 2456.1234567890123, 1028.5, 0.0, 0.0, 1.0]
    q = tmp_path / "ext_T.xml"
    q.write_text(text.replace("<rows>3</rows>", "<rows>3</rows>").replace("<cols>3</cols>", "<cols>1</cols>").replace("<dt>d</dt>", "<dt>f</dt>")
                 .replace("2.4560000000000000e+03 0. 1.2275000000000000e+03 0.", "-2.50000000e+00 3.99999991e-02").replace("2.4561234567890123e+03 1.0285000000000000e+03 0. 0. 1.", "-1.25000000e-01"))
    r = subprocess.run([dump, "xml", str(q)], capture_output=True, text=True)
    lines = r.stdout.split()
    assert lines[:2] == ["3", "1"] and [float(v) for v in lines[2:]] == [-2.5, 0.0399999991, -0.125]
    bad = tmp_path / "bad.xml"
    bad.write_text(text.replace("0. 0. 1.</data>", "</data>"))                      # six values for a 3 x 3 matrix
    r = subprocess.run([dump, "xml", str(bad)], capture_output=True, text=True)
    assert "matrix data truncated" in r.stdout and r.stdout.strip().endswith("0 0")    # an empty matrix: load_data then reports "invalid intrinsics"
