"""The C++ sequence driver wass_stereo_batch (replacement for wasscli's fan-out, cli/wasscli/wasscli.py:305-364).

CPU: the driver's own logic -- sharding over worker processes, the gather through pipes, planes.txt bytes and the
NaN-aware mean -- on workdirs that are already finished (--skip-existing reads plane.txt back), against the golden
planes_txt.npz (numpy nanmean semantics of wassgridsurface.py:672-678).
GPU: real frames through persistent contexts, compared with one wass_stereo process per frame; Coll-1 over RCCL.
"""
import os
import shutil
import subprocess

import numpy as np
import pytest

from test_cli import make_workdir

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="module")
def exes():
    from wass_amd import build
    cli = build.build_host()
    return cli, build.BATCH


def _finished_workdir(root, i, line):
    wd = os.path.join(root, "%06d_wd" % i)
    os.makedirs(wd)
    with open(os.path.join(wd, "plane.txt"), "w") as f:
        f.write(line + "\n" if "nan" in line else "".join(v + "\n" for v in line.split()))
    open(os.path.join(wd, "mesh_cam.xyzC"), "wb").write(b"\0" * 148)
    return wd


def test_usage(exes):
    r = subprocess.run([exes[1]], capture_output=True, text=True)
    assert r.returncode == 0 and "Usage:" in r.stdout


@pytest.mark.parametrize("workers", [1, 2, 3])
def test_planes_txt_and_mean_from_finished_workdirs(exes, tmp_path, workers):
    g = np.load(os.path.join(GOLD, "planes_txt.npz"))
    lines = str(g["text"]).strip().split("\n")
    out = tmp_path / "output"
    out.mkdir()
    for i, l in enumerate(lines):
        _finished_workdir(str(out), i, l)
    cfg = tmp_path / "cfg.txt"
    cfg.write_text("MAX_DISPARITY=64\n")
    r = subprocess.run([exes[1], str(cfg), "--sequence", str(out), "--gpus", "1", "--procs-per-gpu", str(workers), "--skip-existing"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert (out / "planes.txt").read_text() == str(g["text"])          # byte for byte, frame order
    mean = np.array([float(x) for x in (out / "planes_mean.txt").read_text().split()])
    np.testing.assert_allclose(mean, g["nanmean"], rtol=1e-15)
    assert f"mean plane over {int(g['n_valid'])} frame(s)" in r.stdout
    for i in range(len(lines)):
        assert f"[frame {i}]" in r.stdout


def test_long_sequences_do_not_fill_the_worker_pipes(exes, tmp_path):
    """More records per worker than a 64 KiB pipe holds (~1000 of 64 bytes): the parent has to drain every worker as it
    goes -- draining them one after the other blocks workers 1.. in write() and, with an all-reduce before the tail,
    deadlocks the run."""
    out = tmp_path / "output"
    out.mkdir()
    n = 2 * 1300
    for i in range(n):
        _finished_workdir(str(out), i, "0.1 0.2 0.3 %d" % i)
    cfg = tmp_path / "cfg.txt"
    cfg.write_text("MAX_DISPARITY=64\n")
    r = subprocess.run([exes[1], str(cfg), "--sequence", str(out), "--gpus", "1", "--procs-per-gpu", "2", "--skip-existing"],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr
    lines = (out / "planes.txt").read_text().strip().split("\n")
    assert len(lines) == n and lines[0].split()[3] == "0" and lines[-1].split()[3] == str(n - 1)
    mean = np.array([float(x) for x in (out / "planes_mean.txt").read_text().split()])
    np.testing.assert_allclose(mean, [0.1, 0.2, 0.3, (n - 1) / 2], rtol=1e-12)


def test_failed_frame_is_reported_and_left_out(exes, tmp_path):
    out = tmp_path / "output"
    out.mkdir()
    _finished_workdir(str(out), 0, "0.1 0.2 0.3 -4")
    os.makedirs(out / "000001_wd")                                       # unfinished and unprocessable (no inputs, no GPU needed to fail)
    _finished_workdir(str(out), 2, "0.3 0.2 0.1 -6")
    cfg = tmp_path / "cfg.txt"
    cfg.write_text("MAX_DISPARITY=64\n")
    r = subprocess.run([exes[1], str(cfg), "--sequence", str(out), "--procs-per-gpu", "2", "--skip-existing"], capture_output=True, text=True)
    assert r.returncode == 255
    assert (out / "planes.txt").read_text() == "0.1 0.2 0.3 -4\n0.3 0.2 0.1 -6\n"
    assert "[frame 1]" in r.stdout and "rc=-1" in r.stdout
    np.testing.assert_allclose([float(x) for x in (out / "planes_mean.txt").read_text().split()], [0.2, 0.2, 0.2, -5.0], rtol=1e-15)


def test_without_rccl_the_parent_reduces_the_planes_itself(exes, tmp_path):
    """--rccl-always on a box where librccl cannot be loaded (WASS_RCCL_LIB points nowhere and the default names are hidden by an empty
    LD_LIBRARY_PATH on the build container; on a GPU box this test finds RCCL and is skipped): the sequence is not lost -- round 5 printed
    "planes will be reduced by the parent process" and then gave up."""
    out = tmp_path / "output"
    out.mkdir()
    for i in range(4):
        _finished_workdir(str(out), i, "0.1 0.2 0.3 %d" % i)
    cfg = tmp_path / "cfg.txt"
    cfg.write_text("MAX_DISPARITY=64\n")
    env = dict(os.environ, WASS_RCCL_LIB=str(tmp_path / "nowhere.so"))
    r = subprocess.run([exes[1], str(cfg), "--sequence", str(out), "--gpus", "1", "--rccl-always", "--skip-existing"], capture_output=True, text=True, env=env)
    if "RCCL is not available" not in r.stderr:
        assert r.returncode == 0 and "(RCCL all-reduce)" in r.stdout      # a box with a loadable librccl: the leg ran with a world of one
    else:
        assert r.returncode == 0, r.stdout + r.stderr
        assert "(RCCL all-reduce)" not in r.stdout
    np.testing.assert_allclose([float(x) for x in (out / "planes_mean.txt").read_text().split()], [0.1, 0.2, 0.3, 1.5], rtol=1e-15)


@pytest.mark.gpu
def test_a_worker_without_pytorch_in_its_process_finds_rccl_and_reduces_with_a_world_of_one(exes, tmp_path):
    """Row e readiness (no multi-GPU node has ever been available): `bench.py` and the Python tests run RCCL inside a process that has
    PyTorch's copy loaded.  The shipped workers do not.  --rccl-always makes ONE worker run the leg the N-GPU run depends on -- the
    parent takes the unique id (librccl by dlopen), forks the worker, the worker builds a communicator of one rank from it on its own
    context's stream and all-reduces the five doubles -- so the library search, the fork pattern and the call sequence are tested on a
    1-GPU box."""
    cli, batch = exes
    w, h, D = 320, 240, 64
    seq = tmp_path / "seq"
    cfg = None
    for i in range(3):
        t = tmp_path / f"mk{i}"
        t.mkdir()
        wd, cfg, *_ = make_workdir(str(t), w, h, D, frame=i)
        shutil.copytree(wd, seq / ("%06d_wd" % i))
    env = {k: v for k, v in os.environ.items() if k != "WASS_RCCL_LIB"}
    r = subprocess.run([batch, cfg, "--sequence", str(seq), "--gpus", "1", "--rccl-always"], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    assert "RCCL is not available" not in r.stderr and "(RCCL all-reduce)" in r.stdout
    planes = np.array([[float(x) for x in l.split()] for l in (seq / "planes.txt").read_text().strip().split("\n")])
    np.testing.assert_allclose([float(x) for x in (seq / "planes_mean.txt").read_text().split()], np.nanmean(planes, axis=0), rtol=1e-14)


@pytest.mark.gpu
@pytest.mark.parametrize("layout", [("--procs-per-gpu", "2"), ("--threads-per-proc", "3"), ("--procs-per-gpu", "2", "--threads-per-proc", "2")])
def test_batch_equals_one_process_per_frame(exes, tmp_path, layout):
    """Worker processes and / or threads (each thread a context of its own inside one process): the files of every workdir are
    those of one wass_stereo process per frame, byte for byte, and the per-frame log goes to the right workdir."""
    cli, batch = exes
    w, h, D = 320, 240, 64
    seq_a, seq_b = tmp_path / "a", tmp_path / "b"
    cfg = None
    nframes = 5
    for i in range(nframes):
        t = tmp_path / f"mk{i}"
        t.mkdir()
        wd, cfg_i, *_ = make_workdir(str(t), w, h, D, frame=i)
        for seq in (seq_a, seq_b):
            shutil.copytree(wd, seq / ("%06d_wd" % i))
        cfg = cfg_i
    for i in range(nframes):                                             # the reference's way: one process per frame
        r = subprocess.run([cli, cfg, str(seq_a / ("%06d_wd" % i))], capture_output=True, text=True)
        assert r.returncode == 0, r.stdout
    r = subprocess.run([batch, cfg, "--sequence", str(seq_b), "--gpus", "1", *layout], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    expect = ""
    for i in range(nframes):
        log = (seq_b / ("%06d_wd" % i) / "wass_stereo_log.txt").read_text()
        assert log.count("Reconstructing") == 1 and ("%06d_wd" % i) in log and "estimated plane coeffs" in log
        # (the previews: resized by load_data on the host in wass_stereo, on the GPU in the driver's pipelined chain)
        for name in ("mesh_cam.xyzC", "plane.txt", "P0cam.txt", "Cam1_poseT.txt", "plane_refinement_inliers.xyz", "00000000_s.png", "00000001_s.png",
                     "K0_small.txt", "scale.txt"):
            a = (seq_a / ("%06d_wd" % i) / name).read_bytes()
            assert a == (seq_b / ("%06d_wd" % i) / name).read_bytes(), f"frame {i}: {name} differs"
        expect += " ".join((seq_a / ("%06d_wd" % i) / "plane.txt").read_text().split("\n")).strip() + "\n"
    assert (seq_b / "planes.txt").read_text() == expect
    planes = np.array([[float(x) for x in l.split()] for l in expect.strip().split("\n")])
    np.testing.assert_allclose([float(x) for x in (seq_b / "planes_mean.txt").read_text().split()], np.nanmean(planes, axis=0), rtol=1e-14)
    # the same sequence again: nothing is recomputed
    r2 = subprocess.run([batch, cfg, "--sequence", str(seq_b), "--skip-existing"], capture_output=True, text=True)
    assert r2.returncode == 0 and (seq_b / "planes.txt").read_text() == expect


@pytest.mark.gpu
def test_every_frame_of_a_pipelined_sequence_has_its_gpu_stage_times(exes, tmp_path):
    """The time table of every frame's log carries GPU event times for the tail stages -- also for the frames that are NOT the last of
    the sequence: the driver enqueues frame n+1's triangulation before it reads frame n's record, so the tail's timing events are two
    sets (round 4 had one, and every frame but the last showed 0 s for Triangulation .. Plane refinement)."""
    import re
    cli, batch = exes
    w, h, D = 320, 240, 64
    seq = tmp_path / "seq"
    cfg = None
    nframes = 4
    for i in range(nframes):
        t = tmp_path / f"mk{i}"
        t.mkdir()
        wd, cfg, *_ = make_workdir(str(t), w, h, D, frame=i)
        shutil.copytree(wd, seq / ("%06d_wd" % i))
    r = subprocess.run([batch, cfg, "--sequence", str(seq), "--gpus", "1"], capture_output=True, text=True)
    assert r.returncode == 0 and "pipelined" in r.stdout, r.stdout + r.stderr
    for i in range(nframes):
        log = (seq / ("%06d_wd" % i) / "wass_stereo_log.txt").read_text()
        rows = dict((m.group(1).strip(), float(m.group(2))) for m in re.finditer(r"\|\s+([A-Za-z][A-Za-z \-]+?)\s+\|\s+([0-9.eE+\-]+) \|", log))
        for stage in ("Dense Stereo", "Triangulation", "Z-gap stats", "Outlier removal", "Plane fitting", "Plane refinement"):
            assert rows.get(stage, 0.0) > 0.0, f"frame {i}: row {stage!r} of the time table is {rows.get(stage)} ({sorted(rows)})"


@pytest.mark.gpu
def test_rccl_allreduce_through_the_c_abi(gpu_ctx):
    """Coll-1 over RCCL on one rank (the multi-rank case needs one GPU per rank): unique id, communicator, all-reduce."""
    import ctypes as C
    from wass_amd import _lib
    lib = _lib.load()
    uid = (C.c_ubyte * 128)()
    assert lib.wass_coll_unique_id(uid) == 0
    assert lib.wass_coll_init(gpu_ctx._h, 0, 1, uid) == 0, gpu_ctx._lib.wass_last_error(gpu_ctx._h).decode()
    acc = (C.c_double * 5)(1.5, -2.0, 3.25, -11.0, 4.0)
    assert lib.wass_coll_allreduce_sum_f64(gpu_ctx._h, acc, 5) == 0, gpu_ctx._lib.wass_last_error(gpu_ctx._h).decode()
    assert list(acc) == [1.5, -2.0, 3.25, -11.0, 4.0]


@pytest.mark.gpu
def test_bench_runs_two_ranks_or_fails_loudly():
    """python bench.py --gpus 2 on a one-GPU box must never be a silent one-rank run."""
    import json
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--config", "A",
                        "--no-cpu-baseline"], capture_output=True, text=True, cwd=ROOT)
    import torch
    if torch.cuda.device_count() < 2:
        assert r.returncode != 0 and "only 1 device(s) are visible" in (r.stdout + r.stderr)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--config", "A",
                        "--no-cpu-baseline", "--allow-shared-gpu"], capture_output=True, text=True, cwd=ROOT)
    assert r.returncode == 0, r.stdout + r.stderr
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["ranks"]["world_size_from_process_group"] == 2
    assert len(line["ranks"]["pairs_per_sec_per_rank"]) == 2 and line["planes_averaged"] == 4


@pytest.mark.gpu
def test_bench_multi_rank_epilogue_with_the_librarys_rccl_leg():
    """Everything `bench.py --gpus N` does after the timed region on RCCL -- the process-group collectives, the sequence mean
    plane through wass_coll_init / wass_coll_allreduce_sum_f64 (the product's own path, cross-checked against torch), the
    gathered per-rank rates -- run with a process group of ONE rank: the code the 8-GPU run executes, on the one GPU there is."""
    import json
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--rccl-single-rank", "--steps", "4", "--warmup", "1", "--config", "A",
                        "--no-cpu-baseline"], capture_output=True, text=True, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    pa = line["plane_allreduce"]
    assert pa is not None and "error" not in pa, pa
    assert pa["path"].startswith("wass_coll_allreduce_sum_f64") and pa["matches_torch_distributed"] is True
    assert line["ranks"]["backend"] == "nccl" and line["planes_averaged"] == 4


def test_raw_mode_prepares_workdirs_and_fails_loudly_without_a_gpu(exes, tmp_path):
    """--raw (prepare-less mode) on a machine without a GPU: the host side still does wass_prepare's part of every workdir
    (calibration copies, wasscli's numbering of the pairs), every frame fails loudly in its own log, and nothing pretends to
    have been computed."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from test_cli import _write_png, _write_xml
    from wass_amd import synth
    w, h = 96, 64
    rig = synth.rig_geometry(w, h)
    calib = tmp_path / "config"; calib.mkdir()
    _write_xml(calib / "intrinsics_00.xml", "intr", rig["K_left"]); _write_xml(calib / "intrinsics_01.xml", "intr", rig["K_right"])
    _write_xml(calib / "ext_R.xml", "R", rig["R"]); _write_xml(calib / "ext_T.xml", "T", np.array(rig["T"]).reshape(3, 1))
    cam0 = tmp_path / "cam0"; cam1 = tmp_path / "cam1"; cam0.mkdir(); cam1.mkdir()
    rng = np.random.default_rng(3)
    for t in range(3):
        _write_png(cam0 / ("b_%03d.png" % t), rng.integers(0, 255, (h, w), dtype=np.uint8))
        _write_png(cam1 / ("a_%03d.png" % t), rng.integers(0, 255, (h, w), dtype=np.uint8))
    (cam0 / "notes.txt").write_text("not a picture")
    cfg = tmp_path / "cfg.txt"; cfg.write_text("MAX_DISPARITY=16\n")
    out = tmp_path / "output"
    r = subprocess.run([exes[1], str(cfg), "--raw", str(calib), "--cam0", str(cam0), "--cam1", str(cam1), "--sequence", str(out), "--frames", "2"],
                       capture_output=True, text=True)
    assert r.returncode == 255 and "2 frame(s)" in r.stdout and "pipelined" in r.stdout, r.stdout + r.stderr
    for t in range(2):
        wd = out / ("%06d_wd" % t)
        assert "rc=-1" in r.stdout and (wd / "intrinsics_00000000.xml").exists() and (wd / "ext_T.xml").exists() and (wd / "stereo_config.txt").exists()
        log = (wd / "wass_stereo_log.txt").read_text()
        assert "no usable MI355X GPU" in log and f"image 0 loaded, Size: {w}x{h}" in log
        assert not (wd / "mesh_cam.xyzC").exists() and not (wd / "undistorted").exists()
    assert not (out / "000002_wd").exists()
    # argument checking: --raw needs both camera directories and extrinsics in the calibration directory
    r = subprocess.run([exes[1], str(cfg), "--raw", str(calib), "--cam0", str(cam0), "--sequence", str(out)], capture_output=True, text=True)
    assert r.returncode == 255 and "--cam0 <dir> --cam1 <dir>" in r.stderr
    (calib / "ext_R.xml").unlink()
    r = subprocess.run([exes[1], str(cfg), "--raw", str(calib), "--cam0", str(cam0), "--cam1", str(cam1), "--sequence", str(out)], capture_output=True, text=True)
    assert r.returncode == 255 and "Extrinsic calibration not found" in r.stderr
