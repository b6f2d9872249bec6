"""The drop-in wass_stereo executable: argv / exit codes / config format / file outputs.

CPU tests cover argv / config handling and the loud failure without a GPU; the GPU tests run --rectify-only and
BASELINE config A (640x480, D=64) end to end through the CLI, with the built-in and with the cv::stereoRectify
rectification, and check the outputs against the oracle chain.
"""
import os
import struct
import subprocess
import zlib

import numpy as np
import pytest

from wass_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def cli():
    from wass_amd import build
    return build.build_host()


def _write_png(path, img):
    h, w = img.shape
    raw = b"".join(b"\x00" + img[y].tobytes() for y in range(h))

    def chunk(t, d):
        c = struct.pack(">I", len(d)) + t + d
        return c + struct.pack(">I", zlib.crc32(t + d) & 0xFFFFFFFF)
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 0, 0, 0, 0)) +
                chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))


def _write_xml(path, node, m):
    m = np.atleast_2d(np.asarray(m, float))
    data = " ".join(repr(float(v)) for v in m.ravel())
    with open(path, "w") as f:
        f.write(f'<?xml version="1.0"?>\n<opencv_storage>\n<{node} type_id="opencv-matrix">\n  <rows>{m.shape[0]}</rows>\n'
                f'  <cols>{m.shape[1]}</cols>\n  <dt>d</dt>\n  <data>\n    {data}</data></{node}>\n</opencv_storage>\n')


def make_workdir(tmp, w, h, D, frame=0, extra_cfg=""):
    """A wass_prepare/wass_match/wass_autocalibrate-style workdir for the ideal rig of SURVEY.md 8(d)."""
    wd = os.path.join(tmp, "000000_wd")
    os.makedirs(os.path.join(wd, "undistorted"))
    right, left = synth.make_pair(w, h, D, frame_idx=frame)
    rig = synth.rig_geometry(w, h)
    _write_png(os.path.join(wd, "undistorted", "00000000.png"), left)     # cam0 = left (T.x > 0: no swap)
    _write_png(os.path.join(wd, "undistorted", "00000001.png"), right)
    _write_xml(os.path.join(wd, "intrinsics_00000000.xml"), "intr", rig["K_left"])
    _write_xml(os.path.join(wd, "intrinsics_00000001.xml"), "intr", rig["K_right"])
    _write_xml(os.path.join(wd, "ext_R.xml"), "R", rig["R"])
    _write_xml(os.path.join(wd, "ext_T.xml"), "T", np.array(rig["T"]).reshape(3, 1) * 2.5)   # 2.5 m baseline -> normalised
    cfg = os.path.join(tmp, "stereo_config.txt")
    with open(cfg, "w") as f:
        f.write(f"# synthetic rig\nMAX_DISPARITY={D}\nRANDOM_SEED=12345\nUSE_CUSTOM_STEREORECTIFY=true\n"
                f"RECTIFY_ANGLE=1e-6\nDISABLE_RECTIFY_ROI=true\n" + extra_cfg)
    return wd, cfg, right, left, rig


def run(cli, *args, cwd=None):
    return subprocess.run([cli, *args], capture_output=True, text=True, cwd=cwd)


# ------------------------------------------------------------------ CPU: boundary behaviour
def test_no_arguments_prints_usage_and_exits_zero(cli):
    r = run(cli)
    assert r.returncode == 0 and "Usage:" in r.stdout and "wass_stereo [--genconfig] <config_file> <workdir>" in r.stdout


def test_invalid_argument_count(cli):
    r = run(cli, "a")
    assert r.returncode == 255 and "Invalid arguments" in r.stderr
    r = run(cli, "a", "b", "c", "d")
    assert r.returncode == 255


def test_missing_workdir(cli, tmp_path):
    r = run(cli, "cfg.txt", str(tmp_path / "nope"))
    assert r.returncode == 255 and "does not exists" in r.stderr


def test_genconfig_format(cli, tmp_path):
    r = run(cli, "--genconfig", cwd=str(tmp_path))
    assert r.returncode == 0
    txt = open(tmp_path / "stereo_config.txt").read()
    blocks = [b for b in txt.split("\n\n") if b.strip()]
    keys = []
    for b in blocks:
        lines = b.split("\n")
        assert len(lines) == 3 and lines[0].startswith("# ") and lines[1] == "# " and lines[2].startswith("#")
        keys.append(lines[2][1:].split("=")[0])
    assert keys == sorted(keys)                                   # alphabetical
    d = dict(b.split("\n")[2][1:].split("=", 1) for b in blocks)
    assert d["MAX_DISPARITY"] == "640" and d["WINSIZE"] == "13" and d["DENSE_SPECKLE_WINDOW_SIZE"] == "-70"
    assert d["LEFT_MASK_IMAGE"] == "none" and d["SAVE_COMPRESSED"] == "true" and d["PLANE_RANSAC_ROUNDS"] == "400"
    assert len(keys) >= 46


def test_bad_config_is_an_error(cli, tmp_path):
    wd, cfg, *_ = make_workdir(str(tmp_path), 64, 48, 16)
    open(cfg, "a").write("NO_SUCH_KEY=1\n")
    r = run(cli, cfg, wd)
    assert r.returncode == 255 and "unknown option NO_SUCH_KEY" in r.stdout
    open(cfg, "w").write("MAX_DISPARITY=abc\n")
    r = run(cli, cfg, wd)
    assert r.returncode == 255 and "invalid value" in r.stdout


@pytest.mark.gpu
def test_rectify_only_is_identity_for_the_ideal_rig(cli, tmp_path):
    w, h, D = 160, 120, 32
    wd, cfg, right, left, rig = make_workdir(str(tmp_path), w, h, D)
    r = run(cli, cfg, wd, "--rectify-only")
    assert r.returncode == 0, r.stdout + r.stderr
    assert "[P|10|100]" in r.stdout and "[P|20|100]" in r.stdout and "All done." in r.stdout
    for f in ("P0cam.txt", "P1cam.txt", "Cam0_poseR.txt", "Cam0_poseT.txt", "Cam1_poseR.txt", "Cam1_poseT.txt", "H0_rect.txt",
              "H1_rect.txt", "stereo_config.txt", "wass_stereo_log.txt", "K0_small.txt", "scale.txt"):
        assert os.path.exists(os.path.join(wd, f)), f
    H0 = np.loadtxt(os.path.join(wd, "H0_rect.txt")); H1 = np.loadtxt(os.path.join(wd, "H1_rect.txt"))
    np.testing.assert_allclose(H0 / H0[2, 2], np.eye(3), atol=1e-6)
    np.testing.assert_allclose(H1 / H1[2, 2], np.eye(3), atol=1e-6)
    T1 = np.loadtxt(os.path.join(wd, "Cam1_poseT.txt"))
    np.testing.assert_allclose(T1, [1, 0, 0], atol=1e-15)         # |T| normalised to 1 (wass_stereo.cpp:360-370)
    P1 = np.loadtxt(os.path.join(wd, "P1cam.txt"))
    np.testing.assert_allclose(P1, rig["K_right"] @ np.hstack([np.eye(3), [[1], [0], [0]]]), rtol=1e-15)
    txt = open(os.path.join(wd, "P0cam.txt")).read()
    assert not txt.endswith("\n") and "e+0" in txt                # scientific, 16 digits, no trailing newline
    assert "load_data [info ] image 0 loaded, Size: 160x120" in open(os.path.join(wd, "wass_stereo_log.txt")).read()


def _read_png_gray(path):
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
    assert (raw[:, 0] == 0).all()                                # filter type 0 on every row
    return raw[:, 1:]


def test_scaled_previews_are_written(cli, tmp_path):
    """SAVE_INPUT_SCALE (default 0.3): 0000000{0,1}_s.png = cv::resize(INTER_CUBIC) of the inputs, K{0,1}_small.txt,
    scale.txt (wass_stereo.cpp:401-434).  Written by load_data, i.e. before the GPU is needed."""
    w, h, D = 200, 150, 32
    wd, cfg, right, left, rig = make_workdir(str(tmp_path), w, h, D)
    run(cli, cfg, wd)                                            # exit code depends on whether a GPU is present
    for name, img in (("00000000_s.png", left), ("00000001_s.png", right)):
        got = _read_png_gray(os.path.join(wd, name)).astype(int)
        assert got.shape == (int(h * 0.3), int(w * 0.3))
        # float64 Keys (a = -0.75) interpolation at the same sample positions, replicated border
        nh, nw = got.shape

        def taps(n_out, n_in):
            f = (np.arange(n_out) + 0.5) * (n_in / n_out) - 0.5
            i0 = np.floor(f).astype(int); t = f - i0
            A = -0.75
            wts = np.stack([((A * (t + 1) - 5 * A) * (t + 1) + 8 * A) * (t + 1) - 4 * A, ((A + 2) * t - (A + 3)) * t * t + 1,
                            ((A + 2) * (1 - t) - (A + 3)) * (1 - t) * (1 - t) + 1], 0)
            wts = np.vstack([wts, 1 - wts.sum(0)])
            idx = np.clip(i0[None, :] + np.arange(-1, 3)[:, None], 0, n_in - 1)
            return idx, wts
        ix, wx = taps(nw, w); iy, wy = taps(nh, h)
        tmp = (img.astype(float)[:, ix] * wx[None]).sum(1)       # [h, nw]
        ref = (tmp[iy] * wy[:, :, None]).sum(0)                  # [nh, nw]
        assert np.abs(got - np.clip(np.rint(ref), 0, 255)).max() <= 1
    assert abs(float(open(os.path.join(wd, "scale.txt")).read()) - 0.3) < 1e-12
    assert os.path.exists(os.path.join(wd, "K0_small.txt")) and os.path.exists(os.path.join(wd, "K1_small.txt"))


def test_no_gpu_is_a_loud_failure(cli, tmp_path):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    wd, cfg, *_ = make_workdir(str(tmp_path), 64, 48, 16)
    r = run(cli, cfg, wd)
    assert r.returncode == 255 and "no usable MI355X GPU" in r.stdout
    assert not os.path.exists(os.path.join(wd, "mesh_cam.xyzC"))
    assert r.stdout.isascii()                                       # wasscli.py:335,339 decodes the output as ASCII


def test_no_gpu_is_a_loud_failure_in_the_pipelined_chain_too(cli, tmp_path):
    """WASS_DEBUG_IMAGES=0 sends the frame through the device-resident chain (frame_pipeline.hpp): same loud failure, same
    host-side files (they are written before the GPU is needed), a log in the workdir."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    wd, cfg, *_ = make_workdir(str(tmp_path), 64, 48, 16)
    r = subprocess.run([cli, cfg, wd], capture_output=True, text=True, env=dict(os.environ, WASS_DEBUG_IMAGES="0"))
    assert r.returncode == 255 and "no usable MI355X GPU" in r.stdout and "[P|10|100]" in r.stdout
    assert not os.path.exists(os.path.join(wd, "mesh_cam.xyzC"))
    log = open(os.path.join(wd, "wass_stereo_log.txt")).read()
    assert "no usable MI355X GPU" in log and "load_data [info ] image 0 loaded, Size: 64x48" in log and "[P|" not in log
    for name in ("P0cam.txt", "Cam1_poseT.txt", "H0_rect.txt", "stereo_config.txt", "K0_small.txt", "scale.txt", "00000000_s.png"):
        assert os.path.exists(os.path.join(wd, name)), name


# ------------------------------------------------------------------ GPU: BASELINE config A through the CLI
@pytest.mark.gpu
@pytest.mark.parametrize("extra", ["", "USE_CUSTOM_STEREORECTIFY=false\n", "DENSE_PATHS=8\nMEDIAN_FILTER_WSIZE=3\nDISCARD_BURNED_AREAS=false\n",
                                   "LEFT_MASK_IMAGE=lmask.png\nRIGHT_MASK_IMAGE=rmask.png\n", "RIGHT_MASK_IMAGE=missing.png\nDISCARD_BURNED_AREAS=false\n"])
def test_pipelined_chain_writes_the_files_of_the_stage_by_stage_calls(cli, tmp_path, extra):
    """wass_stereo runs a frame through the host-sync-free device chain (frame_pipeline.hpp) -- the debug pictures are drawn
    from maps fetched afterwards -- or, on request (WASS_STAGE_BY_STAGE=1) and for a few options, through the synchronous
    per-stage calls: every file must come out the same, byte for byte, the debug pictures included, and the log must carry the
    same numbers."""
    import shutil
    w, h, D = 400, 300, 64
    wd, cfg, *_ = make_workdir(str(tmp_path), w, h, D, extra_cfg=extra)
    if "lmask.png" in extra:                                   # camera masks (wass_stereo.cpp:1059-1087): a ship's bow and a pole
        lm = np.full((h, w), 255, np.uint8); lm[h - 60:, :120] = 0
        rm = np.full((h, w), 255, np.uint8); rm[:, 300:310] = 0; rm[20:40, 50:90] = 0
        _write_png(os.path.join(wd, "lmask.png"), lm)
        _write_png(os.path.join(wd, "rmask.png"), rm)
    wd2 = os.path.join(str(tmp_path), "pipelined_wd")
    shutil.copytree(wd, wd2)
    wd3 = os.path.join(str(tmp_path), "pipelined_nodebug_wd")
    shutil.copytree(wd, wd3)
    a = subprocess.run([cli, cfg, wd], capture_output=True, text=True, env=dict(os.environ, WASS_STAGE_BY_STAGE="1"))
    b = subprocess.run([cli, cfg, wd2], capture_output=True, text=True)
    c = subprocess.run([cli, cfg, wd3], capture_output=True, text=True, env=dict(os.environ, WASS_DEBUG_IMAGES="0"))
    assert a.returncode == 0 and b.returncode == 0 and c.returncode == 0, a.stdout[-2000:] + b.stdout[-2000:] + c.stdout[-1000:]
    assert "pipelined chain:" in b.stdout and "pipelined chain:" in c.stdout and "pipelined chain:" not in a.stdout   # which path ran
    for stage in ("Data load", "Rectification", "Dense Stereo", "Triangulation", "Z-gap stats", "Outlier removal", "Plane fitting",
                  "Plane refinement", "TOTAL"):
        assert stage in b.stdout, stage                           # the reference's time-table rows, GPU times
    pics = ["stereo.jpg", "stereo_input.jpg", "disparity_stereo_ouput.jpg", "disparity_final_scaled.jpg", "disparity_coverage.jpg", "graph_components.jpg",
            "undistorted/R0.jpg", "undistorted/R1.jpg"]
    for name in pics:
        assert open(os.path.join(wd, name), "rb").read() == open(os.path.join(wd2, name), "rb").read(), name
        assert not os.path.exists(os.path.join(wd3, name)), name
    for name in ("mesh_cam.xyzC", "plane.txt", "plane_refinement_inliers.xyz"):
        assert open(os.path.join(wd, name), "rb").read() == open(os.path.join(wd3, name), "rb").read(), name
    for name in ("mesh_cam.xyzC", "plane.txt", "plane_refinement_inliers.xyz", "P0cam.txt", "P1cam.txt", "Cam0_poseR.txt", "Cam1_poseT.txt",
                 "K0_small.txt", "K1_small.txt", "scale.txt", "00000000_s.png", "00000001_s.png", "stereo_config.txt") + \
            (() if "USE_CUSTOM_STEREORECTIFY=false" in extra else ("H0_rect.txt", "H1_rect.txt")):
        assert open(os.path.join(wd, name), "rb").read() == open(os.path.join(wd2, name), "rb").read(), name
    for marker in ("[P|10|100]", "[P|20|100]", "[P|40|100]", "[P|60|100]", "[P|80|100]", "[P|90|100]", "[P|100|100]", "All done."):
        assert marker in b.stdout
    import re
    def numbers(out):                                          # the log lines that carry results
        keep = ("valid points found", "biggest component size", "ransac rounds", "ransac plane coeffs", "refinement inliers",
                "estimated plane coeffs", "number of points after plane cropping", "total data size", "rectification map generated")
        return [l for l in out.splitlines() if any(k in l for k in keep)]
    assert numbers(a.stdout) == numbers(b.stdout) and len(numbers(a.stdout)) == 9
    masks = [l.replace(wd, "WD") for l in a.stdout.splitlines() if "camera mask" in l or "not found or invalid image" in l]
    assert masks == [l.replace(wd2, "WD") for l in b.stdout.splitlines() if "camera mask" in l or "not found or invalid image" in l]
    assert len(masks) == (2 if "MASK_IMAGE" in extra else 0)
    log = open(os.path.join(wd2, "wass_stereo_log.txt")).read()
    assert "[P|" not in log and log.count("Reconstructing") == 1 and "All done." in log


@pytest.mark.gpu
def test_pipelined_chain_too_few_points(cli, tmp_path):
    """Untextured input through the device chain: the point count arrives with the result record, the frame fails like the
    reference (:1993) and leaves no plane.txt / mesh_cam.xyzC behind."""
    wd, cfg, *_ = make_workdir(str(tmp_path), 160, 120, 32)
    flat = np.full((120, 160), 90, np.uint8)
    _write_png(os.path.join(wd, "undistorted", "00000000.png"), flat)
    _write_png(os.path.join(wd, "undistorted", "00000001.png"), flat)
    r = subprocess.run([cli, cfg, wd], capture_output=True, text=True, env=dict(os.environ, WASS_DEBUG_IMAGES="0"))
    assert r.returncode == 255 and "Too few points triangulated" in r.stdout
    assert not os.path.exists(os.path.join(wd, "mesh_cam.xyzC")) and not os.path.exists(os.path.join(wd, "plane.txt"))



@pytest.mark.gpu
def test_config_a_end_to_end_matches_oracle_chain(cli, tmp_path, oracle):
    w, h, D = 640, 480, 64
    wd, cfg, right, left, rig = make_workdir(str(tmp_path), w, h, D, extra_cfg="SAVE_AS_PLY=true\n")
    r = run(cli, cfg, wd)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr
    for marker in ("[P|10|100]", "[P|20|100]", "[P|40|100]", "[P|60|100]", "[P|80|100]", "[P|90|100]", "[P|100|100]", "All done."):
        assert marker in r.stdout
    for stage in ("Data load", "Rectification", "Dense Stereo", "Triangulation", "Z-gap stats", "Outlier removal", "Plane fitting",
                  "Plane refinement", "TOTAL"):
        assert stage in r.stdout
    # oracle chain on the same inputs, using the homographies the CLI wrote
    H0 = np.loadtxt(os.path.join(wd, "H0_rect.txt")); H1 = np.loadtxt(os.path.join(wd, "H1_rect.txt"))
    g = dict(rig); g["HLi"] = np.linalg.inv(H0); g["HRi"] = np.linalg.inv(H1)
    p = oracle.wass_params(D, 5)
    d16, _ = oracle.dense_disparity16(right, left, p)
    f = oracle.disparity_postprocess(d16, 1, D)
    roi = (0, 0, w, h)
    _check_outputs(r, wd, oracle, f, roi, roi, oracle.make_geom(g, use_custom=True), right, left, 0.7)


def _check_outputs(r, wd, oracle, f, roi_l, roi_r, geom, right, left, min_fill):
    """The CLI's log lines and files against the oracle's mesh chain run on the oracle's disparity map f."""
    h, w = right.shape
    full = np.zeros((h, w), np.float32)                         # env.disparity: the ROI map pasted at roi_comb_right (:990)
    full[roi_r[1]:roi_r[1] + roi_r[3], roi_r[0]:roi_r[0] + roi_r[2]] = f
    n, v, p3, gr = oracle.triangulate(full, roi_l, roi_r, geom, right, (left <= 254).astype(np.uint8), (right <= 254).astype(np.uint8))
    zg, _ = oracle.zgap_percentile(v, p3, 99.0)
    v, _ = oracle.keep_biggest_component(v, p3, zg)
    uv = oracle.ransac_sample(roi_r[2], roi_r[3], 400, 12345)
    ok, pl, best, _ = oracle.ransac_plane(v, p3, uv, 1.0)
    assert ok
    v, _ = oracle.crop_plane(v, p3, pl, 1.0)
    pl2, ninl, _ = oracle.refine_plane(v, p3)
    v, _ = oracle.crop_plane(v, p3, pl2, 1.5)

    assert f"{n} valid points found" in r.stdout
    assert f"400 ransac rounds, {best} best inliers" in r.stdout
    plane = np.loadtxt(os.path.join(wd, "plane.txt"))
    assert plane.shape == (4,)
    # tolerance: 1e-8 absolute on a unit normal / d in baseline units.  With the paper's 2.5 m baseline that is
    # 2.5e-5 mm -- three orders of magnitude below the xyzC quantisation step.
    np.testing.assert_allclose(plane, pl2, rtol=0, atol=1e-8)
    blob = open(os.path.join(wd, "mesh_cam.xyzC"), "rb").read()
    npts = struct.unpack("<I", blob[:4])[0]
    assert abs(npts - int(v.sum())) <= 2 and len(blob) == 148 + 6 * npts
    if npts == int(v.sum()):
        ref = oracle.encode_xyzc(v, p3, plane)                    # same plane -> same bytes
        q = np.frombuffer(blob[148:], np.uint16); qr = np.frombuffer(ref[148:], np.uint16)
        assert (q != qr).mean() < 1e-4                            # a last-bit difference in HLi may move a few quanta
    # the recovered surface is the synthetic sea plane: > 70 % of the pixels end up in the cloud
    assert npts > min_fill * w * h
    # mesh.ply header (PovMesh.cpp:473-483)
    head = open(os.path.join(wd, "mesh.ply"), "rb").read(200).decode("latin1")
    assert head.startswith(f"ply\nformat binary_little_endian 1.0\nelement vertex {npts}\nproperty float x\n")
    # plane_refinement_inliers.xyz: "x y z" per line in the stream's default (%g) format (wass_stereo.cpp:2077-2085)
    lines = open(os.path.join(wd, "plane_refinement_inliers.xyz")).read().splitlines()
    assert len(lines) > 100
    for ln in lines[:50] + lines[-50:]:
        toks = ln.split(" ")
        assert len(toks) == 3 and all("%g" % float(t) == t for t in toks), ln


@pytest.mark.gpu
def test_fixed_random_seed_runs_are_reproducible(cli, tmp_path):
    """RANDOM_SEED != -1 (wass_stereo.cpp:1864-1872): two runs over the same workdir write identical files."""
    import shutil
    wd, cfg, *_ = make_workdir(str(tmp_path), 160, 120, 16)
    wd2 = os.path.join(str(tmp_path), "again_wd")
    shutil.copytree(wd, wd2)
    for d in (wd, wd2):
        assert run(cli, cfg, d).returncode == 0
    for name in ("mesh_cam.xyzC", "plane.txt", "plane_refinement_inliers.xyz", "P0cam.txt", "disparity_final_scaled.jpg"):
        assert open(os.path.join(wd, name), "rb").read() == open(os.path.join(wd2, name), "rb").read(), name


@pytest.mark.gpu
def test_config_a_with_opencv_rectification(cli, tmp_path, oracle):
    """USE_CUSTOM_STEREORECTIFY=false (the reference's default): cv::stereoRectify alpha=1 + initUndistortRectifyMap +
    bicubic remap (wass_stereo.cpp:530-610), then the same chain on the cropped ROIs."""
    w, h, D = 640, 480, 64
    wd, cfg, right, left, rig = make_workdir(str(tmp_path), w, h, D, extra_cfg="SAVE_AS_PLY=true\nUSE_CUSTOM_STEREORECTIFY=false\n")
    r = run(cli, cfg, wd)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr
    assert "Rectifying via cv::stereoRectify" in r.stdout and "All done." in r.stdout
    rr = oracle.stereo_rectify(rig["K_left"], rig["K_right"], w, h, rig["R"], rig["T"], 1.0)
    assert rr["P2"][0, 3] >= 0                                                  # no swap for this rig
    rl, rq = rr["roi1"], rr["roi2"]
    ymin = max(rl[1], rq[1]); ymax = min(rl[1] + rl[3], rq[1] + rq[3])
    wmin = min(rl[2], rq[2])
    roi_l = (rl[0], ymin, wmin, ymax - ymin); roi_r = (rq[0], ymin, wmin, ymax - ymin)
    assert f"rectification map generated. Size: {wmin}x{ymax - ymin}" in r.stdout
    lx, ly = oracle.init_rectify_map(rig["K_left"], rr["R1"], rr["P1"], w, h)
    rx, ry = oracle.init_rectify_map(rig["K_right"], rr["R2"], rr["P2"], w, h)
    crop = lambda a, q: np.ascontiguousarray(a[q[1]:q[1] + q[3], q[0]:q[0] + q[2]])  # noqa: E731
    left_crop = crop(oracle.remap_cubic(left, lx, ly), roi_l)
    right_crop = crop(oracle.remap_cubic(right, rx, ry), roi_r)
    d16, _ = oracle.dense_disparity16(right_crop, left_crop, oracle.wass_params(D, 5))
    f = oracle.disparity_postprocess(d16, 1, D)
    g = dict(rig); g.update(R1=rr["R1"], R2=rr["R2"], P1=rr["P1"], P2=rr["P2"])
    _check_outputs(r, wd, oracle, f, roi_l, roi_r, oracle.make_geom(g, use_custom=False), right, left, 0.65)


@pytest.mark.gpu
def test_ransac_failure_writes_nan_plane_and_continues(cli, tmp_path):
    """Untextured input -> (almost) nothing triangulated -> too few points -> exit -1, like the reference (:1993)."""
    wd, cfg, *_ = make_workdir(str(tmp_path), 160, 120, 32)
    flat = np.full((120, 160), 90, np.uint8)
    _write_png(os.path.join(wd, "undistorted", "00000000.png"), flat)
    _write_png(os.path.join(wd, "undistorted", "00000001.png"), flat)
    r = run(cli, cfg, wd)
    assert r.returncode == 255 and "Too few points triangulated" in r.stdout


def _read_png(path):
    b = open(path, "rb").read()
    assert b[:8] == b"\x89PNG\r\n\x1a\n"
    pos, idat, hdr = 8, b"", None
    while pos < len(b):
        n, t = struct.unpack(">I", b[pos:pos + 4])[0], b[pos + 4:pos + 8]
        d = b[pos + 8:pos + 8 + n]
        assert zlib.crc32(t + d) & 0xFFFFFFFF == struct.unpack(">I", b[pos + 8 + n:pos + 12 + n])[0]
        if t == b"IHDR":
            hdr = struct.unpack(">IIBBBBB", d)
        elif t == b"IDAT":
            idat += d
        pos += 12 + n
    w, h, depth, ctype = hdr[:4]
    ch = {0: 1, 2: 3}[ctype]
    raw = np.frombuffer(zlib.decompress(idat), np.uint8).reshape(h, w * ch + 1)
    assert depth == 8 and (raw[:, 0] == 0).all()
    return raw[:, 1:].reshape(h, w, ch).squeeze()


@pytest.mark.gpu
def test_debug_pictures_of_the_reference_are_written(cli, tmp_path, oracle):
    """stereo / stereo_input / disparity_stereo_ouput (sic) / disparity_final_scaled /
    disparity_coverage / graph_components, same stems as the reference (PNG instead of JPEG), same pixel arithmetic."""
    w, h, D = 320, 240, 64
    wd, cfg, right, left, rig = make_workdir(str(tmp_path), w, h, D)
    # WASS_DEBUG_FORMAT=png: the same pictures, lossless, so that the pixel arithmetic can be checked exactly
    r = subprocess.run([cli, cfg, wd], capture_output=True, text=True, env=dict(os.environ, WASS_DEBUG_FORMAT="png"))
    assert r.returncode == 0, r.stdout
    st = _read_png(os.path.join(wd, "stereo.png"))
    assert st.shape == (h, 2 * w, 3) and (st[0, :, 0] == 255).all() and (st[20, :, 1] == 0).all()       # a red line every 20 rows
    np.testing.assert_array_equal(st[7, 10:w - 10, 1], left[7, 10:w - 10])                               # left | right, grey
    np.testing.assert_array_equal(st[7, w + 10:2 * w - 10, 2], right[7, 10:w - 10])
    si = _read_png(os.path.join(wd, "stereo_input.png"))
    assert si.shape == (2 * h, w + D) and (si[:, :D] == 0).all()
    np.testing.assert_array_equal(si[:h, D:], left); np.testing.assert_array_equal(si[h:, D:], right)
    # disparity pictures: (v - min) / (max - min) * 255 of the oracle's maps (render.hpp:101-136)
    p = oracle.wass_params(D)
    d16, _ = oracle.dense_disparity16(right, left, p)
    final = oracle.disparity_postprocess(d16, 1, D)
    fs = _read_png(os.path.join(wd, "disparity_final_scaled.png"))
    mn, mx = np.float32(min(final.min(), w + 1)), np.float32(final.max())
    np.testing.assert_array_equal(fs, ((final - mn) / (mx - mn) * np.float32(255.0)).astype(np.uint8))
    raw = _read_png(os.path.join(wd, "disparity_stereo_ouput.png"))
    assert raw.shape == (h, w) and ((raw > 0) >= (fs > 0)).mean() > 0.99          # clean-up only ever removes pixels (up to hole filling)
    cov = _read_png(os.path.join(wd, "disparity_coverage.png"))
    assert cov.shape == (h // 2, w // 2, 3) and (cov[h // 4, w // 4] == [right[h // 2:h // 2 + 2, w // 2:w // 2 + 2].astype(int).sum() + 2 >> 2, 100,
                                                                         right[h // 2:h // 2 + 2, w // 2:w // 2 + 2].astype(int).sum() + 2 >> 2]).all()
    # undistorted/R0, R1 (wass_stereo.cpp:1381-1382): grey of the rectified right / matched left pixel where a point was made
    R0 = _read_png(os.path.join(wd, "undistorted", "R0.png")); R1 = _read_png(os.path.join(wd, "undistorted", "R1.png"))
    assert R0.shape == (h, w, 3) and R1.shape == (h, w, 3)
    grey0 = (R0[..., 0] == R0[..., 1]) & (R0[..., 1] == R0[..., 2]) & (R0.sum(-1) > 0)
    assert grey0.mean() > 0.5 and (R0[grey0][:, 0] == right[grey0]).mean() > 0.99     # identity rectification: R0's grey IS the right image
    grey1 = (R1[..., 0] == R1[..., 1]) & (R1[..., 1] == R1[..., 2]) & (R1.sum(-1) > 0)
    assert grey1.mean() > 0.5
    gc = _read_png(os.path.join(wd, "graph_components.png"))
    assert gc.shape == (h // 2, w // 2, 3) and (gc[..., 1] == 255).mean() > 0.5 and (gc[..., 0] == 0).all()
    # default: the reference's file names (.jpg, cv::imwrite), decodable by an independent decoder, the same pictures
    wd3, cfg3, *_ = make_workdir(str(tmp_path / "c"), w, h, D)
    r3 = run(cli, cfg3, wd3)
    assert r3.returncode == 0, r3.stdout
    Image = pytest.importorskip("PIL.Image")
    for name in ("stereo", "stereo_input", "disparity_stereo_ouput", "disparity_final_scaled", "disparity_coverage", "graph_components",
                 os.path.join("undistorted", "R0"), os.path.join("undistorted", "R1")):
        assert not os.path.exists(os.path.join(wd3, name + ".png"))
        jpg = np.asarray(Image.open(os.path.join(wd3, name + ".jpg"))).astype(int)
        png = _read_png(os.path.join(wd, name + ".png")).astype(int)
        assert jpg.shape == png.shape, name
        assert np.abs(jpg - png).mean() < 4.0, (name, np.abs(jpg - png).mean())
    # WASS_DEBUG_IMAGES=0 switches them off
    wd2, cfg2, *_ = make_workdir(str(tmp_path / "b"), w, h, D)
    r2 = subprocess.run([cli, cfg2, wd2], capture_output=True, text=True, env=dict(os.environ, WASS_DEBUG_IMAGES="0"))
    assert r2.returncode == 0 and not os.path.exists(os.path.join(wd2, "stereo.png")) and not os.path.exists(os.path.join(wd2, "stereo.jpg"))
    assert open(os.path.join(wd2, "mesh_cam.xyzC"), "rb").read() == open(os.path.join(wd, "mesh_cam.xyzC"), "rb").read()
