"""Row f2: wass_prepare -- CLAHE (oracle known answers on CPU, HIP parity on the GPU) and the drop-in executable
(reference: src/wass_prepare/wass_prepare.cpp:36-39,257-275,303-540)."""
import os
import subprocess

import numpy as np
import pytest

from test_cli import _read_png_gray, _write_png, _write_xml


@pytest.fixture(scope="module")
def prepare():
    from wass_amd import build
    build.build_host()
    return build.PREPARE


def run(exe, *args, cwd=None):
    return subprocess.run([exe, *args], capture_output=True, text=True, cwd=cwd)


# ------------------------------------------------------------------ CPU: the oracle's CLAHE against hand-computed answers
def test_clahe_flat_histogram_is_plain_equalisation(oracle):
    """One tile holding every grey level equally often and no clipping: LUT[v] = round((v + 1) * 255 / 256)."""
    img = np.arange(256, dtype=np.uint8).reshape(16, 16)
    got = oracle.clahe(img, 0.0, 1)
    v = img.astype(np.float32)
    np.testing.assert_array_equal(got, np.rint((v + 1) * (np.float32(255) / np.float32(256))).astype(np.uint8))


def test_clahe_constant_image_clip_and_redistribution(oracle):
    """A constant 32 x 32 tile (area 1024), clip limit 2: clip = int(2 * 1024 / 256) = 8; the 1016 clipped counts give
    3 to every bin plus one more to bins 0, 1, ..., 247 (remainder 248, step 1).  The bin of the constant v = 100 ends
    at 8 + 3 + 1 = 12; cumulative(100) = 100 * 4 + 12 = 412 -> LUT = rint(412 * 255 / 1024) = 103."""
    img = np.full((32, 32), 100, np.uint8)
    got = oracle.clahe(img, 2.0, 1)
    assert (got == 103).all()
    # remainder with a step > 1: area 1024, clip limit 3.5 -> clip 14, clipped 1010 = 3 * 256 + 242 ... step 1 again;
    # a 40 x 40 tile (area 1600), clip limit 1 -> clip 6, clipped 1594 = 6 * 256 + 58 -> step 4: bins 0, 4, ..., 228 get one more.
    img = np.full((40, 40), 100, np.uint8)
    got = oracle.clahe(img, 1.0, 1)
    cum = 100 * 6 + 26 + (6 + 6 + 1)        # bins 0..99: 6 each + 25 extras (0, 4, ..., 96); bin 100: clip 6 + 6 + 1 extra (100 % 4 == 0)
    assert (got == int(np.rint(np.float32(cum) * (np.float32(255) / np.float32(1600))))).all()


def test_clahe_blend_and_padding_properties(oracle):
    rng = np.random.default_rng(5)
    # one tile -> every pixel uses the same table: the map is monotone in the input grey level
    img = rng.integers(0, 256, (48, 64), dtype=np.uint8)
    out = oracle.clahe(img, 4.0, 1)
    order = np.argsort(img.ravel(), kind="stable")
    assert (np.diff(out.ravel()[order].astype(int)) >= 0).all()
    # a grid that does not divide the image: the result keeps the input size and equals the result for the
    # REFLECT_101-extended image cropped back (tile size of the extended image = that of the padded one)
    img = rng.integers(0, 256, (50, 70), dtype=np.uint8)
    out = oracle.clahe(img, 2.0, 8)
    assert out.shape == img.shape
    pad = np.pad(img, ((0, 8 - 50 % 8), (0, 8 - 70 % 8)), mode="reflect")
    np.testing.assert_array_equal(oracle.clahe(pad, 2.0, 8)[:50, :70], out)
    # horizontally mirrored input -> mirrored output is NOT guaranteed (tile coordinates are asymmetric), but a
    # constant image stays constant whatever the grid
    flat = np.full((50, 70), 17, np.uint8)
    assert len(np.unique(oracle.clahe(flat, 2.0, 5))) == 1


# ------------------------------------------------------------------ CPU: executable boundary
def test_prepare_no_arguments_prints_options_and_exits_zero(prepare):
    r = run(prepare)
    assert r.returncode == 0
    for opt in ("--workdir", "--calibdir", "--c0", "--c1", "--demosaic", "--hdr", "--dolp-aolp", "--save-channels", "--save-stokes",
                "--continue-if-existing", "--genconfig"):
        assert opt in r.stdout


def test_prepare_genconfig(prepare, tmp_path):
    r = run(prepare, "--genconfig", cwd=str(tmp_path))
    assert r.returncode == 0
    text = open(tmp_path / "prepare_config.txt").read()
    assert text.startswith("# CAM0 CLAHE cliplimit parameter\n# \n#CAM0_CLAHE_CLIPLIMIT=2.0\n\n")
    assert [l[1:].split("=")[0] for l in text.splitlines() if l.startswith("#CAM")] == [
        "CAM0_CLAHE_CLIPLIMIT", "CAM0_CLAHE_TILEGRIDSIZE", "CAM1_CLAHE_CLIPLIMIT", "CAM1_CLAHE_TILEGRIDSIZE"]


def test_prepare_argument_errors(prepare, tmp_path):
    calib = tmp_path / "calib"; calib.mkdir()
    wd = str(tmp_path / "wd")
    cases = [
        (["--bogus"], "unrecognised option"),
        (["--calibdir", str(calib), "--c0", "a.png", "--c1", "b.png"], "workdir option not specified"),
        (["--workdir", wd, "--c0", "a.png", "--c1", "b.png"], "calibdir option not specified"),
        (["--workdir", wd, "--calibdir", str(calib), "--c0", "a.png"], "c0 and c1 options must be both specified"),
        (["--workdir", wd, "--calibdir", str(tmp_path / "nope"), "--c0", "a.png", "--c1", "b.png"], "Invalid calibration directory"),
        (["--workdir", str(calib), "--calibdir", str(calib), "--c0", "a.png", "--c1", "b.png"], "already exists."),
        (["--workdir", wd, "--calibdir", str(calib), "--c0", "a.png", "--c1", "b.png", "--demosaic"], "polarimetric"),
    ]
    for args, msg in cases:
        r = run(prepare, *args)
        assert r.returncode == 255 and msg in r.stdout, (args, r.stdout)
    assert not os.path.exists(wd)
    # missing intrinsics: the workdir is created (as in the reference) and the program stops with -1
    r = run(prepare, "--workdir", wd, "--calibdir", str(calib), "--c0", "a.png", "--c1", "b.png")
    assert r.returncode == 255 and "[P|10|100]" in r.stdout and "Unable to load" in r.stdout and os.path.isdir(wd)
    # an ill-typed configuration value is an error before anything is created
    (calib / "prepare_config.txt").write_text("CAM0_CLAHE_TILEGRIDSIZE=many\n")
    r = run(prepare, "--workdir", str(tmp_path / "wd2"), "--calibdir", str(calib), "--c0", "a.png", "--c1", "b.png")
    assert r.returncode == 255 and "invalid value" in r.stdout and not os.path.exists(tmp_path / "wd2")


def _calibdir(tmp_path, w, h, clahe_cfg=None, distortion=True):
    calib = tmp_path / "calib"; calib.mkdir()
    K0 = np.array([[0.9 * w, 0, w / 2 - 3.5], [0, 0.9 * w, h / 2 + 2.25], [0, 0, 1]])
    K1 = np.array([[0.92 * w, 0, w / 2 + 1.5], [0, 0.92 * w, h / 2 - 4.0], [0, 0, 1]])
    d0 = np.array([-0.21, 0.08, 1e-3, -5e-4, 0.01])
    d1 = np.array([-0.18, 0.05, -7e-4, 3e-4, 0.0])
    _write_xml(calib / "intrinsics_00.xml", "intr", K0)
    _write_xml(calib / "intrinsics_01.xml", "intr", K1)
    if distortion:
        _write_xml(calib / "distortion_00.xml", "dist", d0.reshape(5, 1))
        _write_xml(calib / "distortion_01.xml", "dist", d1.reshape(1, 5))      # row or column vector, as calibration tools write either
    R = np.eye(3); T = np.array([[2.5], [0.01], [-0.02]])
    _write_xml(calib / "ext_R.xml", "R", R)
    _write_xml(calib / "ext_T.xml", "T", T)
    if clahe_cfg is not None:
        (calib / "prepare_config.txt").write_text(clahe_cfg)
    return calib, K0, K1, (d0 if distortion else np.zeros(5)), (d1 if distortion else np.zeros(5)), R, T


def _images(tmp_path, w, h, seed=3):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    out = []
    for k in range(2):
        img = 90 + 50 * np.sin(xx / (7.0 + k)) * np.cos(yy / (5.0 + 2 * k)) + rng.normal(0, 9, (h, w))
        img = np.clip(img, 0, 255).astype(np.uint8)
        _write_png(tmp_path / f"cam{k}.png", img)
        out.append(img)
    return out


def test_prepare_without_gpu_is_a_loud_failure(prepare, tmp_path):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    calib, *_ = _calibdir(tmp_path, 96, 64)
    _images(tmp_path, 96, 64)
    wd = tmp_path / "wd"
    r = run(prepare, "--workdir", str(wd), "--calibdir", str(calib), "--c0", str(tmp_path / "cam0.png"), "--c1", str(tmp_path / "cam1.png"))
    assert r.returncode == 255 and "unable to open the GPU" in r.stdout
    assert not os.path.exists(wd / "undistorted" / "00000000.png")
    # the argv wasscli really sends (wasscli.py:226-227): an EMPTY argument where --demosaic would be, then --continue-if-existing --
    # boost::program_options ignores positional tokens the program does not declare, so the reference never sees it
    r = run(prepare, "--workdir", str(wd), "--calibdir", str(calib), "--c0", str(tmp_path / "cam0.png"), "--c1", str(tmp_path / "cam1.png"), "", "--continue-if-existing")
    assert r.returncode == 255 and "unable to open the GPU" in r.stdout and "unrecognised option" not in r.stdout
    assert r.stdout.isascii()                                         # (wasscli decodes the output as ASCII)


# ------------------------------------------------------------------ GPU: HIP CLAHE vs the oracle, bit-exact
@pytest.mark.gpu
@pytest.mark.parametrize("w,h,tiles,clip", [(64, 48, 1, 2.0), (64, 48, 4, 2.0), (70, 50, 8, 2.0), (333, 257, 16, 40.0), (100, 60, 150, 2.0),
                                            (612, 512, 8, 0.0), (2456, 2058, 150, 2.0), (2456, 2058, 8, 3.0), (31, 17, 3, 0.7)])
def test_clahe_matches_oracle(gpu_ctx, oracle, w, h, tiles, clip):
    rng = np.random.default_rng(w * 7 + tiles)
    yy, xx = np.mgrid[0:h, 0:w]
    img = np.clip(100 + 60 * np.sin(xx / 23.0) * np.cos(yy / 17.0) + rng.normal(0, 12, (h, w)), 0, 255).astype(np.uint8)
    if tiles == 16:
        img[:h // 2] = 255                     # saturated half: heavy clipping, large redistribution
    np.testing.assert_array_equal(gpu_ctx.clahe(img, clip, tiles), oracle.clahe(img, clip, tiles))


@pytest.mark.gpu
def test_clahe_rejects_bad_arguments(gpu_ctx):
    img = np.zeros((8, 8), np.uint8)
    with pytest.raises(Exception):
        gpu_ctx.clahe(img, 2.0, 0)


# ------------------------------------------------------------------ GPU: the executable end to end
@pytest.mark.gpu
@pytest.mark.parametrize("clahe_cfg", [None, "CAM0_CLAHE_TILEGRIDSIZE=6\nCAM0_CLAHE_CLIPLIMIT=3.0\nCAM1_CLAHE_TILEGRIDSIZE=150\n"])
def test_prepare_writes_the_workdir_wass_stereo_reads(prepare, tmp_path, oracle, clahe_cfg):
    w, h = 200, 150
    calib, K0, K1, d0, d1, R, T = _calibdir(tmp_path, w, h, clahe_cfg)
    img0, img1 = _images(tmp_path, w, h)
    wd = tmp_path / "out" / "000000_wd"
    r = run(prepare, "--workdir", str(wd), "--calibdir", str(calib), "--c0", str(tmp_path / "cam0.png"), "--c1=" + str(tmp_path / "cam1.png"), "",
            "--continue-if-existing")                                 # (with wasscli's empty argument in --demosaic's place, wasscli.py:226-227)
    assert r.returncode == 0, r.stdout + r.stderr
    marks = [l for l in r.stdout.splitlines() if l.startswith("[P|")]
    assert marks == ["[P|10|100]", "[P|20|100]", "[P|50|100]", "[P|70|100]", "[P|100|100]"]
    assert "All done, exiting" in r.stdout and f"Output image size: {w}x{h}" in r.stdout
    if clahe_cfg is None:
        assert "Unable to load" in r.stdout and "prepare_config.txt" in r.stdout      # logged, not fatal (wass_prepare.cpp:398-401)
        e0, e1 = img0, img1
    else:
        assert "Settings loaded" in r.stdout
        e0, e1 = oracle.clahe(img0, 3.0, 6), oracle.clahe(img1, 2.0, 150)
    np.testing.assert_array_equal(_read_png_gray(wd / "undistorted" / "00000000.png"), oracle.undistort(e0, K0, d0))
    np.testing.assert_array_equal(_read_png_gray(wd / "undistorted" / "00000001.png"), oracle.undistort(e1, K1, d1))
    # the calibration files wass_stereo loads (wass_stereo.cpp:93-124), read back with the same kind of reader
    import re

    def xml(path, node):
        s = open(path).read()
        m = re.search(rf"<{node} type_id=\"opencv-matrix\">\s*<rows>(\d+)</rows>\s*<cols>(\d+)</cols>\s*<dt>d</dt>\s*<data>(.*?)</data>", s, re.S)
        return np.array(m.group(3).split(), float).reshape(int(m.group(1)), int(m.group(2)))
    np.testing.assert_array_equal(xml(wd / "intrinsics_00000000.xml", "intr"), K0)
    np.testing.assert_array_equal(xml(wd / "intrinsics_00000001.xml", "intr"), K1)
    np.testing.assert_array_equal(xml(wd / "ext_R.xml", "R"), R)
    np.testing.assert_array_equal(xml(wd / "ext_T.xml", "T"), T)
    # running again: refused without --continue-if-existing, accepted with it
    args = ["--workdir", str(wd), "--calibdir", str(calib), "--c0", str(tmp_path / "cam0.png"), "--c1", str(tmp_path / "cam1.png")]
    assert run(prepare, *args).returncode == 255
    assert run(prepare, *args, "--continue-if-existing").returncode == 0


@pytest.mark.gpu
def test_prepare_without_distortion_files_copies_the_images(prepare, tmp_path):
    """distortion_0X.xml missing -> zeros(5,1) (wass_prepare.cpp:428-440): cv::undistort with K' = K is the identity."""
    w, h = 96, 64
    calib, *_ = _calibdir(tmp_path, w, h, distortion=False)
    img0, img1 = _images(tmp_path, w, h)
    wd = tmp_path / "wd"
    r = run(prepare, "--workdir", str(wd), "--calibdir", str(calib), "--c0", str(tmp_path / "cam0.png"), "--c1", str(tmp_path / "cam1.png"))
    assert r.returncode == 0 and r.stdout.count("not found. Assuming no distortion.") == 2
    np.testing.assert_array_equal(_read_png_gray(wd / "undistorted" / "00000000.png"), img0)
    np.testing.assert_array_equal(_read_png_gray(wd / "undistorted" / "00000001.png"), img1)


@pytest.mark.gpu
def test_prepare_reads_tiff_inputs(prepare, tmp_path):
    """wasscli hands wass_prepare whatever the cameras wrote: tif / tiff / png (wasscli.py:47); TIFF goes through host/tiff.hpp."""
    Image = pytest.importorskip("PIL.Image")
    w, h = 96, 64
    calib, *_ = _calibdir(tmp_path, w, h, distortion=False)
    img0, img1 = _images(tmp_path, w, h)
    Image.fromarray(img0).save(tmp_path / "cam0.tif", compression="tiff_lzw")
    Image.fromarray(img1).save(tmp_path / "cam1.tiff")
    wd = tmp_path / "wd"
    r = run(prepare, "--workdir", str(wd), "--calibdir", str(calib), "--c0", str(tmp_path / "cam0.tif"), "--c1", str(tmp_path / "cam1.tiff"))
    assert r.returncode == 0, r.stdout
    np.testing.assert_array_equal(_read_png_gray(wd / "undistorted" / "00000000.png"), img0)
    np.testing.assert_array_equal(_read_png_gray(wd / "undistorted" / "00000001.png"), img1)


@pytest.mark.gpu
def test_prepare_then_stereo(prepare, tmp_path):
    """wass_prepare's workdir is accepted by wass_stereo as is (file names, XML node layout, PNG encoding)."""
    from wass_amd import build, synth
    from test_cli import make_workdir
    w, h, D = 160, 120, 16
    wd0, cfg, right, left, rig = make_workdir(str(tmp_path), w, h, D)
    calib = tmp_path / "calib"; calib.mkdir()
    _write_xml(calib / "intrinsics_00.xml", "intr", rig["K_left"])
    _write_xml(calib / "intrinsics_01.xml", "intr", rig["K_right"])
    _write_xml(calib / "ext_R.xml", "R", rig["R"])
    _write_xml(calib / "ext_T.xml", "T", np.array(rig["T"]).reshape(3, 1) * 2.5)
    _write_png(tmp_path / "l.png", left); _write_png(tmp_path / "r.png", right)
    wd = tmp_path / "prepared_wd"
    r = run(prepare, "--workdir", str(wd), "--calibdir", str(calib), "--c0", str(tmp_path / "l.png"), "--c1", str(tmp_path / "r.png"))
    assert r.returncode == 0, r.stdout
    a = subprocess.run([build.CLI, cfg, str(wd)], capture_output=True, text=True)
    b = subprocess.run([build.CLI, cfg, wd0], capture_output=True, text=True)
    assert a.returncode == 0 and b.returncode == 0, a.stdout[-2000:]
    for name in ("mesh_cam.xyzC", "plane.txt", "P0cam.txt", "P1cam.txt"):
        assert open(wd / name, "rb").read() == open(os.path.join(wd0, name), "rb").read(), name


# ------------------------------------------------------------------ GPU: the prepare-less mode of the sequence driver (row f2, fused)
@pytest.mark.gpu
@pytest.mark.parametrize("save_undistorted,swapped", [(False, False), (True, False), (True, True)])
def test_raw_sequence_equals_prepare_then_stereo(prepare, tmp_path, save_undistorted, swapped):
    """wass_stereo_batch --raw runs wass_prepare's undistortion (+ CLAHE) on the GPU INSIDE the frame chain, from the cameras'
    raw pictures: no undistorted/*.png is written or read unless asked for.  Every file of every workdir must equal what the
    two-executable route leaves behind (wass_prepare per frame, then the sequence driver on the prepared workdirs) -- two
    interpolations with the reference's arithmetic, not one fused resampling.  swapped: cam0 is the RIGHT camera (T.x < 0), so
    wass_stereo exchanges the pictures (wass_stereo.cpp:486) -- each must still be undistorted with its own camera's
    coefficients and contrast settings."""
    from wass_amd import build, synth
    w, h, D = 320, 240, 32
    rig = synth.rig_geometry(w, h)
    calib = tmp_path / "config"; calib.mkdir()
    _write_xml(calib / "intrinsics_00.xml", "intr", rig["K_left"])
    _write_xml(calib / "intrinsics_01.xml", "intr", rig["K_right"])
    _write_xml(calib / "distortion_00.xml", "dist", np.array([-0.012, 0.004, 2e-4, -1e-4, 0.0]).reshape(5, 1))
    _write_xml(calib / "distortion_01.xml", "dist", np.array([0.009, -0.003, -1e-4, 2e-4, 1e-3]).reshape(5, 1))
    _write_xml(calib / "ext_R.xml", "R", rig["R"])
    _write_xml(calib / "ext_T.xml", "T", np.array(rig["T"]).reshape(3, 1) * (-2.5 if swapped else 2.5))
    (calib / "prepare_config.txt").write_text("CAM1_CLAHE_TILEGRIDSIZE=4\nCAM1_CLAHE_CLIPLIMIT=40.0\n")
    cfg = tmp_path / "stereo_config.txt"
    cfg.write_text(f"MAX_DISPARITY={D}\nRANDOM_SEED=12345\nUSE_CUSTOM_STEREORECTIFY=true\nRECTIFY_ANGLE=1e-6\nDISABLE_RECTIFY_ROI=true\n")
    cam0 = tmp_path / "input" / "cam0"; cam1 = tmp_path / "input" / "cam1"
    cam0.mkdir(parents=True); cam1.mkdir(parents=True)
    nframes = 3
    for t in range(nframes):
        right, left = synth.make_pair(w, h, D, frame_idx=40 + t)
        _write_png(cam0 / ("%06d_frame.png" % t), right if swapped else left)
        _write_png(cam1 / ("%06d_frame.png" % t), left if swapped else right)
    # route A: wass_prepare per frame (wasscli.py:222-227), then the sequence driver on the prepared workdirs
    seq_a = tmp_path / "a"; seq_a.mkdir()
    for t in range(nframes):
        r = run(prepare, "--workdir", str(seq_a / ("%06d_wd" % t)), "--calibdir", str(calib), "--c0", str(cam0 / ("%06d_frame.png" % t)),
                "--c1", str(cam1 / ("%06d_frame.png" % t)))
        assert r.returncode == 0, r.stdout
    a = subprocess.run([build.BATCH, str(cfg), "--sequence", str(seq_a)], capture_output=True, text=True)
    assert a.returncode == 0 and "pipelined" in a.stdout, a.stdout + a.stderr
    # route B: one command from the raw pictures
    seq_b = tmp_path / "b"
    extra = ["--save-undistorted"] if save_undistorted else []
    b = subprocess.run([build.BATCH, str(cfg), "--raw", str(calib), "--cam0", str(cam0), "--cam1", str(cam1), "--sequence", str(seq_b), *extra],
                       capture_output=True, text=True)
    assert b.returncode == 0, b.stdout + b.stderr
    names = ["mesh_cam.xyzC", "plane.txt", "plane_refinement_inliers.xyz", "P0cam.txt", "P1cam.txt", "Cam0_poseR.txt", "Cam1_poseT.txt", "K0_small.txt",
             "K1_small.txt", "scale.txt", "00000000_s.png", "00000001_s.png", "intrinsics_00000000.xml", "intrinsics_00000001.xml", "ext_R.xml", "ext_T.xml",
             "H0_rect.txt", "H1_rect.txt", "stereo_config.txt"]
    if save_undistorted:
        names += ["undistorted/00000000.png", "undistorted/00000001.png"]
    for t in range(nframes):
        wa, wb = seq_a / ("%06d_wd" % t), seq_b / ("%06d_wd" % t)
        for name in names:
            assert (wa / name).read_bytes() == (wb / name).read_bytes(), f"frame {t}: {name}"
        if not save_undistorted:
            assert not (wb / "undistorted").exists()                  # the PNG round trip is gone, not hidden
        log = (wb / "wass_stereo_log.txt").read_text()
        assert "image 0 loaded, Size: 320x240" in log and "All done." in log
        assert ("auto-swapping left-right images" in log) == swapped
        npts = int.from_bytes((wb / "mesh_cam.xyzC").read_bytes()[:4], "little")
        assert npts > 0.5 * w * h                                     # the distortion is mild: the surface is still recovered
    assert (seq_a / "planes.txt").read_text() == (seq_b / "planes.txt").read_text()


@pytest.mark.gpu
def test_the_call_sequence_of_wasscli_prepare_and_stereo(prepare, tmp_path):
    """What cli/wasscli/wasscli.py does, call for call, from its working directory with RELATIVE paths: find_wass_pipeline (:54-76: every
    program run without arguments must exit 0), do_prepare (:224-234: one wass_prepare per frame with an EMPTY argument where --demosaic
    would be, the output decoded as ASCII), do_stereo (:326-346: `wass_stereo config/stereo_config.txt output/%06d_wd`, four at a time,
    capture_output, plane.txt appended to planes.txt) -- with the per-frame wass_stereo going through the resident worker, debug
    pictures on as the reference has them."""
    import shutil
    from concurrent.futures import ThreadPoolExecutor
    from wass_amd import build, synth
    from test_cli import make_workdir
    w, h, D, N = 160, 120, 16, 6
    root = tmp_path / "work"
    for d in ("config", "input/cam0", "input/cam1", "output"):
        (root / d).mkdir(parents=True)
    sock = tmp_path / "sock"
    sock.mkdir()
    env = dict(os.environ, WASS_SERVER_DIR=str(sock), WASS_SERVER_IDLE="3", PATH=os.path.dirname(build.CLI) + os.pathsep + os.environ.get("PATH", ""))
    env.pop("WASS_NO_SERVER", None)
    env.pop("WASS_DEBUG_IMAGES", None)
    for name in ("wass_prepare", "wass_stereo"):                                   # find_wass_pipeline
        exe = shutil.which(name, path=env["PATH"])
        assert exe and os.path.dirname(exe) == os.path.dirname(build.CLI)
        assert subprocess.run(exe, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, env=env).returncode == 0
    rig = None
    for t in range(N):
        mk = tmp_path / f"mk{t}"
        mk.mkdir()
        wd0, cfg, right, left, rig = make_workdir(str(mk), w, h, D, frame=t)
        _write_png(root / "input" / "cam0" / ("%06d_cam0.png" % t), left)
        _write_png(root / "input" / "cam1" / ("%06d_cam1.png" % t), right)
    shutil.copy(cfg, root / "config" / "stereo_config.txt")
    _write_xml(root / "config" / "intrinsics_00.xml", "intr", rig["K_left"])
    _write_xml(root / "config" / "intrinsics_01.xml", "intr", rig["K_right"])
    _write_xml(root / "config" / "ext_R.xml", "R", rig["R"])                       # (what wass_autocalibrate leaves in config/)
    _write_xml(root / "config" / "ext_T.xml", "T", np.array(rig["T"]).reshape(3, 1) * 2.5)
    for t in range(N):                                                             # do_prepare
        ret = subprocess.run(["wass_prepare", "--workdir", "output/%06d_wd" % t, "--calibdir", "config/", "--c0", "input/cam0/%06d_cam0.png" % t,
                              "--c1", "input/cam1/%06d_cam1.png" % t, "%s" % "", "--continue-if-existing"], capture_output=True, cwd=root, env=env)
        assert ret.returncode == 0, ret.stdout.decode("ascii")
        ret.stdout.decode("ascii")

    planes = []

    def stereo_task(t):                                                            # do_stereo's _stereo_task
        wdirname = "output/%06d_wd" % t
        ret = subprocess.run(["wass_stereo", "config/stereo_config.txt", wdirname], capture_output=True, cwd=root, env=env)
        assert ret.returncode == 0, ret.stdout.decode("ascii")
        text = ret.stdout.decode("ascii")
        assert "[P|100|100]" in text and wdirname in text
        with open(root / wdirname / "plane.txt") as f:
            planes.append((t, (" ".join(f.readlines()).replace("\n", "")) + "\n"))
        return True
    with ThreadPoolExecutor(4) as ex:
        assert all(ex.map(stereo_task, range(N)))
    assert len(planes) == N and all(len(p.split()) == 4 for _, p in planes)
    for t in range(N):
        for f in ("mesh_cam.xyzC", "plane_refinement_inliers.xyz", "stereo.jpg", "undistorted/R0.jpg", "graph_components.jpg", "00000000_s.png", "wass_stereo_log.txt"):
            assert os.path.getsize(root / "output" / ("%06d_wd" % t) / f) > 100, (t, f)
    assert not (root / "stereo_config.txt").exists() and not (root / "wass_stereo_log.txt").exists()     # nothing lands in the caller's directory
    out = subprocess.run(["ps", "-ww", "-eo", "pid,args"], capture_output=True, text=True).stdout
    assert sum(1 for l in out.splitlines() if "--server" in l and str(sock) in l) == 1                      # one worker served all of it
