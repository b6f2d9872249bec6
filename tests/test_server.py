"""The resident worker behind the unchanged command line (wass_amd/host/stereo_server.hpp).

wasscli starts one `wass_stereo <config> <workdir>` process per frame, four at a time (cli/wasscli/wasscli.py:326-346).  Here such
a process hands its frame to a per-GPU server that the first caller starts: same files, same stdout, same exit code.
CPU: the mechanics (start on demand, concurrent callers, relayed log and exit code, idle time-out, WASS_NO_SERVER).
GPU: 4 concurrent callers over 32 workdirs -- what thread_map(..., max_workers=4) does -- against one process per frame in-process."""
import os
import shutil
import subprocess
import time
from concurrent.futures import ThreadPoolExecutor

import pytest

from test_cli import make_workdir

NAMES = ("mesh_cam.xyzC", "plane.txt", "plane_refinement_inliers.xyz", "P0cam.txt", "P1cam.txt", "Cam0_poseR.txt", "Cam1_poseT.txt", "K0_small.txt",
         "K1_small.txt", "scale.txt", "00000000_s.png", "00000001_s.png", "stereo_config.txt")


@pytest.fixture(scope="module")
def cli():
    from wass_amd import build
    return build.build_host()


def _env(sockdir, **kw):
    e = dict(os.environ, WASS_SERVER_DIR=str(sockdir), WASS_SERVER_IDLE="2", **kw)
    e.pop("WASS_NO_SERVER", None)
    return e


def _servers(sockdir):
    out = subprocess.run(["ps", "-ww", "-eo", "pid,args"], capture_output=True, text=True).stdout
    return [l for l in out.splitlines() if "--server" in l and str(sockdir) in l]


def _wait_gone(sockdir, seconds=15):
    t0 = time.time()
    while _servers(sockdir) and time.time() - t0 < seconds:
        time.sleep(0.2)
    return not _servers(sockdir)


def test_server_is_started_on_demand_relays_log_and_exit_code_and_goes_away(cli, tmp_path):
    import torch
    if torch.cuda.is_available():
        pytest.skip("the no-GPU behaviour is tested on the build container")
    wd, cfg, *_ = make_workdir(str(tmp_path), 160, 120, 32)
    sock = tmp_path / "sock"
    sock.mkdir()
    a = subprocess.run([cli, cfg, wd], capture_output=True, text=True, env=_env(sock, WASS_DEBUG_IMAGES="0"))
    assert len(_servers(sock)) == 1                                   # started by the first caller, still there
    b = subprocess.run([cli, cfg, wd], capture_output=True, text=True, env=_env(sock, WASS_DEBUG_IMAGES="0"))
    ref = subprocess.run([cli, cfg, wd], capture_output=True, text=True, env=dict(os.environ, WASS_NO_SERVER="1", WASS_DEBUG_IMAGES="0"))
    for r in (a, b):
        assert r.returncode == ref.returncode == 255                  # no GPU: a loud failure, whoever computes the frame
        assert "no usable MI355X GPU" in r.stdout and "wass_stereo  v." in r.stdout
        assert "[P|10|100]" in r.stdout                               # the markers that were reached come through
    assert os.path.exists(os.path.join(wd, "wass_stereo_log.txt"))
    assert _wait_gone(sock), "the server outlived its idle time-out"
    assert not [f for f in os.listdir(sock) if f.endswith(".sock")]   # and took its socket with it


def test_concurrent_callers_share_one_server(cli, tmp_path):
    import torch
    if torch.cuda.is_available():
        pytest.skip("the no-GPU behaviour is tested on the build container")
    wds = []
    for i in range(6):
        t = tmp_path / f"mk{i}"
        t.mkdir()
        wd, cfg, *_ = make_workdir(str(t), 160, 120, 32, frame=i)
        wds.append((wd, cfg))
    sock = tmp_path / "sock"
    sock.mkdir()
    with ThreadPoolExecutor(4) as ex:
        res = list(ex.map(lambda wc: subprocess.run([cli, wc[1], wc[0]], capture_output=True, text=True, env=_env(sock, WASS_DEBUG_IMAGES="0")), wds))
    assert all(r.returncode == 255 and "Reconstructing" in r.stdout for r in res)
    for (wd, _), r in zip(wds, res):
        assert wd in r.stdout                                         # every caller got ITS frame's log
    assert len(_servers(sock)) <= 1
    _wait_gone(sock)


def test_a_killed_server_costs_time_never_a_frame(cli, tmp_path):
    """The server is killed while idle (its socket file stays behind, nobody listens): the next caller finds the stale socket,
    takes the lock, starts a fresh server and gets its answer; a caller that cannot start one computes the frame itself."""
    import signal
    import torch
    if torch.cuda.is_available():
        pytest.skip("the no-GPU behaviour is tested on the build container")
    wd, cfg, *_ = make_workdir(str(tmp_path), 160, 120, 32)
    sock = tmp_path / "sock"
    sock.mkdir()
    a = subprocess.run([cli, cfg, wd], capture_output=True, text=True, env=_env(sock, WASS_DEBUG_IMAGES="0"))
    srv = _servers(sock)
    assert len(srv) == 1
    os.kill(int(srv[0].split()[0]), signal.SIGKILL)
    time.sleep(0.3)
    assert [f for f in os.listdir(sock) if f.endswith(".sock")]                 # the stale socket is still there
    b = subprocess.run([cli, cfg, wd], capture_output=True, text=True, env=_env(sock, WASS_DEBUG_IMAGES="0"))
    assert b.returncode == a.returncode == 255 and "Reconstructing" in b.stdout    # answered (by a new server)
    assert len(_servers(sock)) == 1
    _wait_gone(sock)
    # a socket "directory" that is a plain file: neither lock nor socket can be created, the caller computes (here: fails loudly) in-process
    notdir = tmp_path / "notdir"
    notdir.write_text("x")
    c = subprocess.run([cli, cfg, wd], capture_output=True, text=True, env=_env(notdir, WASS_DEBUG_IMAGES="0"))
    assert c.returncode == 255 and "no usable MI355X GPU" in c.stdout
    assert not _servers(notdir)


def test_no_server_switch_and_ineligible_configurations_stay_in_process(cli, tmp_path):
    wd, cfg, *_ = make_workdir(str(tmp_path), 160, 120, 32)
    sock = tmp_path / "sock"
    sock.mkdir()
    subprocess.run([cli, cfg, wd], capture_output=True, text=True, env=dict(_env(sock), WASS_NO_SERVER="1"))
    assert not os.listdir(sock)
    subprocess.run([cli, cfg, wd, "--rectify-only"], capture_output=True, text=True, env=_env(sock))
    assert not os.listdir(sock)
    open(cfg, "a").write("SAVE_AS_PLY=true\n")                        # needs the whole mesh on the host: not the chain's business
    subprocess.run([cli, cfg, wd], capture_output=True, text=True, env=_env(sock))
    assert not os.listdir(sock)


@pytest.mark.gpu
def test_four_concurrent_callers_over_32_workdirs_write_the_files_of_in_process_runs(cli, tmp_path):
    w, h, D = 320, 240, 64
    nd, nrep = 8, 4
    seq_a, seq_b = tmp_path / "a", tmp_path / "b"
    cfg = None
    for i in range(nd):
        t = tmp_path / f"mk{i}"
        t.mkdir()
        wd, cfg, *_ = make_workdir(str(t), w, h, D, frame=i)
        for rep in range(nrep):
            for seq in (seq_a, seq_b):
                shutil.copytree(wd, seq / ("%06d_wd" % (rep * nd + i)))
    n = nd * nrep
    sock = tmp_path / "sock"
    sock.mkdir()
    for i in range(nd):                                                  # the reference's way, in-process: one per distinct frame is enough
        r = subprocess.run([cli, cfg, str(seq_a / ("%06d_wd" % i))], capture_output=True, text=True, env=dict(os.environ, WASS_NO_SERVER="1", WASS_DEBUG_IMAGES="0"))
        assert r.returncode == 0, r.stdout[-1500:]
    t0 = time.time()
    with ThreadPoolExecutor(4) as ex:                                    # wasscli: thread_map(..., max_workers=4)
        res = list(ex.map(lambda i: subprocess.run([cli, cfg, str(seq_b / ("%06d_wd" % i))], capture_output=True, text=True,
                                                   env=_env(sock, WASS_DEBUG_IMAGES="0")), range(n)))
    dt = time.time() - t0
    assert len(_servers(sock)) == 1
    for i, r in enumerate(res):
        assert r.returncode == 0, r.stdout[-1500:]
        for marker in ("[P|10|100]", "[P|20|100]", "[P|40|100]", "[P|60|100]", "[P|80|100]", "[P|90|100]", "[P|100|100]", "All done.", "wass_stereo  v."):
            assert marker in r.stdout, (i, marker)
        assert ("%06d_wd" % i) in r.stdout and r.stdout.isascii()     # wasscli.py:335,339: ret.stdout.decode("ascii")
        for name in NAMES + ("H0_rect.txt", "H1_rect.txt"):
            a = (seq_a / ("%06d_wd" % (i % nd)) / name).read_bytes()
            assert a == (seq_b / ("%06d_wd" % i) / name).read_bytes(), f"frame {i}: {name} differs"
    print(f"{n} frames through the server with 4 callers: {n / dt:.1f} frames/s")
    assert _wait_gone(sock)


@pytest.mark.gpu
def test_server_draws_the_debug_pictures_too(cli, tmp_path):
    wd, cfg, *_ = make_workdir(str(tmp_path), 320, 240, 64)
    wd2 = str(tmp_path / "b_wd")
    shutil.copytree(wd, wd2)
    sock = tmp_path / "sock"
    sock.mkdir()
    a = subprocess.run([cli, cfg, wd], capture_output=True, text=True, env=dict(os.environ, WASS_NO_SERVER="1", WASS_DEBUG_FORMAT="png"))
    b = subprocess.run([cli, cfg, wd2], capture_output=True, text=True, env=_env(sock, WASS_DEBUG_FORMAT="png"))
    assert a.returncode == 0 and b.returncode == 0, a.stdout[-800:] + b.stdout[-800:]
    for name in ("stereo.png", "stereo_input.png", "disparity_stereo_ouput.png", "disparity_final_scaled.png", "disparity_coverage.png",
                 "graph_components.png", "undistorted/R0.png", "undistorted/R1.png", "mesh_cam.xyzC", "plane.txt"):
        assert open(os.path.join(wd, name), "rb").read() == open(os.path.join(wd2, name), "rb").read(), name
    _wait_gone(sock)


def test_the_per_frame_client_carries_no_gpu_runtime_and_no_computation(cli, tmp_path):
    """`wass_stereo` (wass_stereo_client.cpp) is what wasscli starts once per frame: it must not map libwassgpu / the HIP runtime (10 ms
    of every call), its banner is the full program's, and without the full program next to it the call fails -- it computes nothing."""
    import ctypes
    from wass_amd import build
    assert cli == build.CLI and os.path.exists(build.CLI_GPU)
    needed = subprocess.run(["readelf", "-d", build.CLI], capture_output=True, text=True).stdout
    assert "NEEDED" in needed
    for lib in ("wassgpu", "amdhip", "hsa-runtime", "libz"):
        assert lib not in needed
    assert "wassgpu" in subprocess.run(["readelf", "-d", build.CLI_GPU], capture_output=True, text=True).stdout
    # one banner, whoever prints it
    lib = ctypes.CDLL(build.SO)
    lib.wass_version.restype = ctypes.c_char_p
    src = open(os.path.join(os.path.dirname(build.CLI), "..", "host", "stereo_client.hpp")).read()
    assert '#define WASS_GPU_LIBRARY_VERSION "%s"' % lib.wass_version().decode() in src
    wd, cfg, *_ = make_workdir(str(tmp_path), 160, 120, 32)
    env = dict(os.environ, WASS_NO_SERVER="1", WASS_DEBUG_IMAGES="0")
    a = subprocess.run([build.CLI, cfg, wd], capture_output=True, text=True, env=env)
    b = subprocess.run([build.CLI_GPU, cfg, wd], capture_output=True, text=True, env=env)
    assert a.returncode == b.returncode and a.stdout.count("wass_stereo  v.") == b.stdout.count("wass_stereo  v.") == 1
    strip = lambda s: [l for l in s.splitlines() if "secs" not in l and "Total" not in l]
    assert strip(a.stdout)[:6] == strip(b.stdout)[:6]
    # alone in a directory: nothing to hand the frame to
    alone = tmp_path / "bin"
    alone.mkdir()
    shutil.copy(build.CLI, alone / "wass_stereo")
    c = subprocess.run([str(alone / "wass_stereo"), cfg, wd], capture_output=True, text=True, env=env)
    assert c.returncode == 255 and "cannot start" in c.stderr and "wass_stereo_gpu" in c.stderr
    assert not os.path.exists(os.path.join(wd, "mesh_cam.xyzC"))


def test_read_ahead_decodes_the_next_workdirs_and_never_serves_a_changed_file(cli, tmp_path):
    """wasscli walks 000000_wd, 000001_wd, ... in order; after two requests from one sequence directory the server inflates the PNGs of
    the next workdirs before their callers exist (ReadAhead, stereo_server.hpp).  A pair decoded early is used only while both files
    are still the ones that were read: a picture replaced in between is decoded again."""
    import numpy as np
    import torch
    from test_cli import _write_png
    if torch.cuda.is_available():
        pytest.skip("the no-GPU behaviour is tested on the build container")
    mk = tmp_path / "mk"
    mk.mkdir()
    wd, cfg, *_ = make_workdir(str(mk), 160, 120, 32)
    seq = tmp_path / "seq"
    seq.mkdir()
    for i in range(6):
        shutil.copytree(wd, seq / ("%06d_wd" % i))
    sock = tmp_path / "sock"
    sock.mkdir()
    tlog = tmp_path / "timing.log"
    env = _env(sock, WASS_DEBUG_IMAGES="0", WASS_SERVER_TIMING=str(tlog), WASS_SERVER_SPECULATE="0")     # (decode only; speculation has its own test)
    call = lambda i: subprocess.run([cli, cfg, str(seq / ("%06d_wd" % i))], capture_output=True, text=True, env=env)
    for i in (0, 1):
        assert "image 0 loaded, Size: 160x120" in call(i).stdout
    time.sleep(0.5)                                                   # 2, 3, 4, 5 are decoded by now
    big = np.zeros((150, 200), np.uint8)
    for k in (0, 1):
        _write_png(str(seq / "000002_wd" / "undistorted" / ("0000000%d.png" % k)), big)      # replaced after it was read
    r2 = call(2)
    r3 = subprocess.run([cli, cfg, str(seq / "000003_wd") + "/"], capture_output=True, text=True, env=env)     # (matlab/run_wass.m spells it with a slash)
    assert "image 0 loaded, Size: 200x150" in r2.stdout and "image 1 loaded, Size: 200x150" in r2.stdout
    assert "image 0 loaded, Size: 160x120" in r3.stdout and "%06d_wd" % 3 in r3.stdout
    _wait_gone(sock)
    rows = {l.split()[0].rstrip("/").rsplit("/", 1)[1]: l.split()[1] for l in tlog.read_text().splitlines() if " total " in l}
    assert rows["000000_wd"] == rows["000001_wd"] == "demand"          # nothing is decoded before a sequence shows
    assert rows["000002_wd"] == "demand" and rows["000003_wd"] == "ahead"
    assert "1 decoded early and changed since" in tlog.read_text()
    # switched off: everything on demand
    tlog.unlink()
    for i in range(4):
        call_off = subprocess.run([cli, cfg, str(seq / ("%06d_wd" % i))], capture_output=True, text=True, env=dict(env, WASS_SERVER_READAHEAD="0"))
        assert "image 0 loaded" in call_off.stdout
    _wait_gone(sock)
    assert " ahead " not in tlog.read_text()


def test_on_a_multi_gpu_node_frame_i_goes_to_gpu_i_mod_g_and_each_server_reads_its_own_stride_ahead(cli, tmp_path):
    """Callers pick the GPU by the number in the workdir's name (the sequence driver's rule, frame i -> GPU i mod G): wasscli's
    consecutive frames land on different GPUs, and the server of GPU g sees g, g+G, g+2G, ... -- which is what it reads ahead."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("the no-GPU behaviour is tested on the build container")
    mk = tmp_path / "mk"
    mk.mkdir()
    wd, cfg, *_ = make_workdir(str(mk), 160, 120, 32)
    seq = tmp_path / "seq"
    seq.mkdir()
    for i in range(8):
        shutil.copytree(wd, seq / ("%06d_wd" % i))
    sock = tmp_path / "sock"
    sock.mkdir()
    tlog = tmp_path / "timing.log"
    env = _env(sock, WASS_DEBUG_IMAGES="0", WASS_SERVER_TIMING=str(tlog), WASS_NUM_GPUS="2", WASS_SERVER_SPECULATE="0")
    env.pop("WASS_GPU_DEVICE", None)
    for i in range(8):
        r = subprocess.run([cli, cfg, str(seq / ("%06d_wd" % i))], capture_output=True, text=True, env=env)
        assert "%06d_wd" % i in r.stdout
        if i == 1:
            assert sorted(f for f in os.listdir(sock) if f.endswith(".sock")) == ["wass_stereo_%d_gpu0.sock" % os.getuid(), "wass_stereo_%d_gpu1.sock" % os.getuid()]
        time.sleep(0.15)
    assert len(_servers(sock)) == 2
    _wait_gone(sock)
    rows = {l.split()[0].rsplit("/", 1)[1]: l.split()[1] for l in tlog.read_text().splitlines() if " total " in l}
    # each server: two requests on demand (0, 2 / 1, 3), then its own stride ahead (4, 6 / 5, 7)
    assert [rows["%06d_wd" % i] for i in range(8)] == ["demand"] * 4 + ["ahead"] * 4
    assert tlog.read_text().count("2 frames decoded before they were asked for, 2 on demand, 0 decoded early") == 2
    # a workdir without a number, or a pinned caller: still served
    plain = tmp_path / "plain"
    shutil.copytree(wd, plain)
    r = subprocess.run([cli, cfg, str(plain)], capture_output=True, text=True, env=env)
    assert "Reconstructing" in r.stdout
    r = subprocess.run([cli, cfg, str(seq / "000003_wd")], capture_output=True, text=True, env=dict(env, WASS_GPU_DEVICE="0"))
    assert "000003_wd" in r.stdout
    _wait_gone(sock)


def test_visibility_variables_and_foreign_parts_limit_the_gpus_a_caller_counts(cli, tmp_path):
    """Round-5 advisor: the GPU count came from the KFD topology alone -- on an 8-GPU node restricted to one device seven of eight callers
    asked for a GPU their server could not open.  The count is now what a server process of THIS environment could use: gfx950 nodes only
    (no APU / iGPU / CPU nodes), at most as many as HIP_VISIBLE_DEVICES & co. list."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("the no-GPU behaviour is tested on the build container")
    topo = tmp_path / "nodes"
    for i, (simd, target) in enumerate([(0, 0)] + [(1024, 90500)] * 4 + [(8, 100306)]):      # a CPU node, four MI355X, an iGPU
        (topo / str(i)).mkdir(parents=True)
        (topo / str(i) / "properties").write_text(f"cpu_cores_count {0 if simd else 64}\nsimd_count {simd}\ngfx_target_version {target}\n")
    mk = tmp_path / "mk"
    mk.mkdir()
    wd, cfg, *_ = make_workdir(str(mk), 160, 120, 32)
    seq = tmp_path / "seq"
    seq.mkdir()
    for i in range(6):
        shutil.copytree(wd, seq / ("%06d_wd" % i))

    def sockets(extra):
        sock = tmp_path / ("sock%d" % len(os.listdir(tmp_path)))
        sock.mkdir()
        env = _env(sock, WASS_DEBUG_IMAGES="0", WASS_KFD_NODES=str(topo), WASS_SERVER_SPECULATE="0", WASS_SERVER_READAHEAD="0", **extra)
        for v in ("WASS_GPU_DEVICE", "WASS_NUM_GPUS", "HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES", "GPU_DEVICE_ORDINAL"):
            if v not in extra:
                env.pop(v, None)
        for i in range(6):
            r = subprocess.run([cli, cfg, str(seq / ("%06d_wd" % i))], capture_output=True, text=True, env=env)
            assert "%06d_wd" % i in r.stdout
        names = sorted(f for f in os.listdir(sock) if f.endswith(".sock"))
        _wait_gone(sock)
        return [n.split("_gpu")[1][:-5] for n in names]
    assert sockets({}) == ["0", "1", "2", "3"]                          # the four gfx950 nodes, not six
    assert sockets({"HIP_VISIBLE_DEVICES": "2"}) == ["0"]               # one visible device: the runtime calls it 0
    assert sockets({"ROCR_VISIBLE_DEVICES": "0,3"}) == ["0", "1"]
    assert sockets({"WASS_NUM_GPUS": "3", "HIP_VISIBLE_DEVICES": "0"}) == ["0", "1", "2"]      # the override is an override


def test_a_configuration_beyond_the_servers_limit_is_computed_by_its_caller(cli, tmp_path):
    """Every distinct configuration text is a pipeline with a context and gigabytes of scratch of its own; the server holds at most
    WASS_SERVER_MAX_CONFIGS (4) of them and REFUSES a further one -- that caller computes its frame in-process -- instead of running
    the GPU out of memory for everybody (round-5 advisor)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("the no-GPU behaviour is tested on the build container")
    wd, cfg, *_ = make_workdir(str(tmp_path), 160, 120, 32)
    cfg2 = str(tmp_path / "cfg2.txt")
    open(cfg2, "w").write(open(cfg).read() + "DENSE_UNIQUENESS_RATIO=5\n")
    sock = tmp_path / "sock"
    sock.mkdir()
    tlog = tmp_path / "timing.log"
    env = _env(sock, WASS_DEBUG_IMAGES="0", WASS_SERVER_MAX_CONFIGS="1", WASS_SERVER_TIMING=str(tlog))
    a = subprocess.run([cli, cfg, wd], capture_output=True, text=True, env=env)
    b = subprocess.run([cli, cfg2, wd], capture_output=True, text=True, env=env)
    c = subprocess.run([cli, cfg, wd], capture_output=True, text=True, env=env)
    for r in (a, b, c):
        assert r.returncode == 255 and "no usable MI355X GPU" in r.stdout and r.stdout.count("wass_stereo  v.") == 1
    _wait_gone(sock)
    assert len([l for l in tlog.read_text().splitlines() if " total " in l]) == 2       # a and c went through the server, b did not


@pytest.mark.gpu
def test_two_sequences_of_different_picture_size_share_one_server(cli, tmp_path):
    """Round-5 advisor (high): the server keeps up to three frames STAGED in front of the GPU; a frame of another picture size used to
    re-allocate the input buffers underneath them (same configuration text = same pipeline).  Two sequences of different cameras, four
    callers, interleaved: every file equals the in-process run's."""
    sizes = ((320, 240), (272, 200))
    nd = 6
    seqs = []
    cfg = None
    for s, (w, h) in enumerate(sizes):
        a, b = tmp_path / f"a{s}", tmp_path / f"b{s}"
        for i in range(nd):
            t = tmp_path / f"mk{s}_{i}"
            t.mkdir()
            wd, c, *_ = make_workdir(str(t), w, h, 64, frame=i)
            if cfg is None:
                cfg = c
            else:
                assert open(c).read() == open(cfg).read()                 # one configuration text: one pipeline in the server
            for seq in (a, b):
                shutil.copytree(wd, seq / ("%06d_wd" % i))
        seqs.append((a, b))
    for a, _ in seqs:
        for i in range(nd):
            r = subprocess.run([cli, cfg, str(a / ("%06d_wd" % i))], capture_output=True, text=True, env=dict(os.environ, WASS_NO_SERVER="1", WASS_DEBUG_IMAGES="0"))
            assert r.returncode == 0, r.stdout[-1500:]
    sock = tmp_path / "sock"
    sock.mkdir()
    jobs = [str(seqs[i % 2][1] / ("%06d_wd" % (i // 2))) for i in range(2 * nd)]           # sizes alternate
    for rnd in range(2):                                                                   # (second round: speculated frames of both sizes in the cache)
        with ThreadPoolExecutor(4) as ex:
            res = list(ex.map(lambda wd: subprocess.run([cli, cfg, wd], capture_output=True, text=True, env=_env(sock, WASS_DEBUG_IMAGES="0")), jobs))
        assert len(_servers(sock)) == 1
        for wd, r in zip(jobs, res):
            assert r.returncode == 0, r.stdout[-1500:]
            ref = wd.replace("/b0/", "/a0/").replace("/b1/", "/a1/")
            for name in NAMES:
                assert open(os.path.join(ref, name), "rb").read() == open(os.path.join(wd, name), "rb").read(), (rnd, wd, name)
    _wait_gone(sock)


@pytest.mark.gpu
def test_a_frame_sent_to_a_gpu_that_is_not_there_is_computed_by_its_caller(cli, tmp_path):
    """Round-5 advisor: a caller picks GPU = frame number mod (GPUs it counts); where the count is too high (WASS_NUM_GPUS, or a topology that
    shows more than the runtime gives this process) the server of the missing device used to ANSWER "no usable GPU" and the frame was lost.
    It now refuses, and the caller computes the frame itself: on this one-GPU box with WASS_NUM_GPUS=2 every second frame takes that route."""
    w, h, D = 320, 240, 64
    wd, cfg, *_ = make_workdir(str(tmp_path), w, h, D)
    seq = tmp_path / "seq"
    for i in range(4):
        shutil.copytree(wd, seq / ("%06d_wd" % i))
    ref = subprocess.run([cli, cfg, wd], capture_output=True, text=True, env=dict(os.environ, WASS_NO_SERVER="1", WASS_DEBUG_IMAGES="0"))
    assert ref.returncode == 0
    sock = tmp_path / "sock"
    sock.mkdir()
    env = _env(sock, WASS_DEBUG_IMAGES="0", WASS_NUM_GPUS="2")
    env.pop("WASS_GPU_DEVICE", None)
    for i in range(4):
        r = subprocess.run([cli, cfg, str(seq / ("%06d_wd" % i))], capture_output=True, text=True, env=env)
        assert r.returncode == 0 and "All done." in r.stdout and r.stdout.count("wass_stereo  v.") == 1, (i, r.stdout[-800:])
        for name in NAMES:
            assert open(os.path.join(wd, name), "rb").read() == (seq / ("%06d_wd" % i) / name).read_bytes(), (i, name)
    assert sorted(f for f in os.listdir(sock) if f.endswith(".sock")) == ["wass_stereo_%d_gpu0.sock" % os.getuid(), "wass_stereo_%d_gpu1.sock" % os.getuid()]
    _wait_gone(sock)


@pytest.mark.gpu
def test_the_server_applies_each_callers_options_not_its_own_environment(cli, tmp_path):
    """Round-5 advisor: WASS_DEBUG_FORMAT / WASS_HOST_INLIER_TEXT were forwarded and keyed but read from the SERVER's environment, i.e.
    from whichever caller had started it.  A server started without them serves a caller that sets them, and the other way round."""
    wd, cfg, *_ = make_workdir(str(tmp_path), 320, 240, 64)
    wds = [str(tmp_path / ("w%d_wd" % k)) for k in range(3)]
    for d in wds:
        shutil.copytree(wd, d)
    sock = tmp_path / "sock"
    sock.mkdir()
    plain, png = _env(sock), _env(sock, WASS_DEBUG_FORMAT="png", WASS_HOST_INLIER_TEXT="1")
    for e in (plain, png):
        for v in ("WASS_DEBUG_IMAGES",):
            e.pop(v, None)
    a = subprocess.run([cli, cfg, wds[0]], capture_output=True, text=True, env=plain)       # starts the server: its environment has neither
    b = subprocess.run([cli, cfg, wds[1]], capture_output=True, text=True, env=png)
    c = subprocess.run([cli, cfg, wds[2]], capture_output=True, text=True, env=plain)
    assert a.returncode == b.returncode == c.returncode == 0, a.stdout[-600:] + b.stdout[-600:]
    assert len(_servers(sock)) == 1
    for d, ext, other in ((wds[0], "jpg", "png"), (wds[1], "png", "jpg"), (wds[2], "jpg", "png")):
        assert os.path.exists(os.path.join(d, "stereo." + ext)) and not os.path.exists(os.path.join(d, "stereo." + other)), (d, ext)
    ref = subprocess.run([cli, cfg, wd], capture_output=True, text=True, env=dict(os.environ, WASS_NO_SERVER="1", WASS_DEBUG_FORMAT="png", WASS_HOST_INLIER_TEXT="1"))
    assert ref.returncode == 0
    for name in ("stereo.png", "disparity_final_scaled.png", "graph_components.png", "plane_refinement_inliers.xyz", "mesh_cam.xyzC"):
        assert open(os.path.join(wd, name), "rb").read() == open(os.path.join(wds[1], name), "rb").read(), name
    _wait_gone(sock)


def test_the_server_survives_garbage_on_its_socket(cli, tmp_path):
    """Anything may connect to a unix socket in /tmp: a wrong magic, a truncated request, an oversized length or a caller that hangs
    up early costs that connection only -- the next real caller is served by the same server."""
    import socket
    import struct
    import torch
    if torch.cuda.is_available():
        pytest.skip("the no-GPU behaviour is tested on the build container")
    wd, cfg, *_ = make_workdir(str(tmp_path), 160, 120, 32)
    sock = tmp_path / "sock"
    sock.mkdir()
    env = dict(_env(sock, WASS_DEBUG_IMAGES="0"), WASS_SERVER_IDLE="4")
    a = subprocess.run([cli, cfg, wd], capture_output=True, text=True, env=env)
    assert "Reconstructing" in a.stdout
    srv = _servers(sock)
    assert len(srv) == 1
    path = str(sock / ("wass_stereo_%d_gpu0.sock" % os.getuid()))
    s4 = lambda b: struct.pack("I", len(b)) + b
    for payload in (b"", b"GET / HTTP/1.0\r\n\r\n", b"WSRV1\n", b"WSRV1\n" + struct.pack("I", 4) + s4(b"cfg") + b"\xff\xff\xff\x7f",
                    b"WSRV1\n" + struct.pack("I", 9), b"WSRV1\n" + struct.pack("I", 4) + s4(b"x") + s4(b"not a configuration") + s4(b"/nonexistent") + s4(b"debug=0"),
                    os.urandom(4096)):
        c = socket.socket(socket.AF_UNIX)
        c.settimeout(5)
        c.connect(path)
        try:
            c.sendall(payload)
            c.shutdown(socket.SHUT_WR)
            c.recv(65536)                                             # a refusal, an error log or nothing -- never a hang
        except OSError:
            pass
        c.close()
    # peers that connect and say nothing: every decode thread gets one, and lets go of it after the receive time-out
    mute = []
    for _ in range(8):
        c = socket.socket(socket.AF_UNIX)
        c.connect(path)
        mute.append(c)
    t0 = time.time()
    b = subprocess.run([cli, cfg, wd], capture_output=True, text=True, env=env, timeout=60)
    assert "Reconstructing" in b.stdout and b.returncode == a.returncode and time.time() - t0 < 20
    for c in mute:
        c.close()
    assert [l.split()[0] for l in _servers(sock)] == [srv[0].split()[0]]      # the SAME server, still alive
    _wait_gone(sock)


def _seq(tmp_path, n, w=160, h=120, D=32):
    mk = tmp_path / "mk"
    mk.mkdir()
    wd, cfg, *_ = make_workdir(str(mk), w, h, D)
    seq = tmp_path / "seq"
    seq.mkdir()
    for i in range(n):
        shutil.copytree(wd, seq / ("%06d_wd" % i))
    return seq, cfg


def test_speculation_prepares_ahead_without_touching_the_workdir(cli, tmp_path):
    """After two requests from a sequence the worker builds (and, with a GPU, computes) the next workdirs' frames before anybody asks:
    until the caller comes NOTHING is written into such a workdir; then it holds what an ordinary call leaves.  A frame whose input
    changed in between is recomputed, one whose caller never comes is dropped."""
    import numpy as np
    import torch
    from test_cli import _write_png
    if torch.cuda.is_available():
        pytest.skip("the no-GPU behaviour is tested on the build container")
    seq, cfg = _seq(tmp_path, 7)
    sock = tmp_path / "sock"
    sock.mkdir()
    tlog = tmp_path / "timing.log"
    env = _env(sock, WASS_DEBUG_IMAGES="0", WASS_SERVER_TIMING=str(tlog))
    call = lambda i: subprocess.run([cli, cfg, str(seq / ("%06d_wd" % i))], capture_output=True, text=True, env=env)
    before = {i: sorted(os.listdir(seq / ("%06d_wd" % i))) for i in range(7)}
    a0, a1 = call(0), call(1)
    time.sleep(0.6)                                                   # 2, 3, 4 are prepared (and have failed on the missing GPU) by now
    for i in (2, 3, 4, 5):
        assert sorted(os.listdir(seq / ("%06d_wd" % i))) == before[i], i            # untouched
    _write_png(str(seq / "000003_wd" / "undistorted" / "00000000.png"), np.zeros((150, 200), np.uint8))
    _write_png(str(seq / "000003_wd" / "undistorted" / "00000001.png"), np.zeros((150, 200), np.uint8))
    r2, r3 = call(2), call(3)
    for r in (a0, a1, r2):
        assert r.returncode == 255 and "no usable MI355X GPU" in r.stdout and "image 0 loaded, Size: 160x120" in r.stdout
    assert "image 0 loaded, Size: 200x150" in r3.stdout               # the frame computed ahead was for other pictures: done again
    for i in (0, 2):
        left = sorted(os.listdir(seq / ("%06d_wd" % i)))
        assert "stereo_config.txt" in left and "P0cam.txt" in left and "wass_stereo_log.txt" in left and "scale.txt" in left
    assert sorted(os.listdir(seq / "000000_wd")) == sorted(os.listdir(seq / "000002_wd"))      # ahead or on demand: the same files
    for name in ("stereo_config.txt", "P0cam.txt", "Cam1_poseT.txt", "K0_small.txt", "scale.txt", "H0_rect.txt"):
        assert (seq / "000000_wd" / name).read_bytes() == (seq / "000002_wd" / name).read_bytes(), name
    assert sorted(os.listdir(seq / "000006_wd")) == before[6]         # never asked for: never touched
    _wait_gone(sock)
    rows = {l.split()[0].rstrip("/").rsplit("/", 1)[1]: l.split()[1] for l in tlog.read_text().splitlines() if " total " in l}
    assert rows["000000_wd"] == rows["000001_wd"] == "demand" and rows["000002_wd"] == "computed" and rows["000003_wd"] == "demand"
    assert sorted(os.listdir(seq / "000006_wd")) == before[6] and sorted(os.listdir(seq / "000005_wd")) == before[5]
    text = tlog.read_text()
    assert "speculation: 1 frames computed before they were asked for" in text and "dropped unclaimed" in text


@pytest.mark.gpu
def test_frames_computed_ahead_of_their_callers_are_the_frames_of_in_process_runs(cli, tmp_path):
    """Speculation with a GPU: a slow caller finds most of its frames computed before it asks; what it gets is byte for byte what a process
    of its own writes -- also for a workdir whose pictures were replaced after the worker had computed the old ones."""
    import numpy as np
    from test_cli import _write_png
    w, h, D = 320, 240, 64
    nd, nrep = 4, 5
    seq_a, seq_b = tmp_path / "a", tmp_path / "b"
    cfg = None
    for i in range(nd):
        t = tmp_path / f"mk{i}"
        t.mkdir()
        wd, cfg, *_ = make_workdir(str(t), w, h, D, frame=i)
        shutil.copytree(wd, seq_a / ("%06d_wd" % i))
        for rep in range(nrep):
            shutil.copytree(wd, seq_b / ("%06d_wd" % (rep * nd + i)))
    n = nd * nrep
    for i in range(nd):
        r = subprocess.run([cli, cfg, str(seq_a / ("%06d_wd" % i))], capture_output=True, text=True, env=dict(os.environ, WASS_NO_SERVER="1", WASS_DEBUG_IMAGES="0"))
        assert r.returncode == 0, r.stdout[-1500:]
    sock = tmp_path / "sock"
    sock.mkdir()
    tlog = tmp_path / "timing.log"
    env = _env(sock, WASS_DEBUG_IMAGES="0", WASS_SERVER_TIMING=str(tlog))
    swapped = 9                                                          # gets frame 2's pictures while the worker holds a result for frame 1's
    for i in range(n):
        if i == 6:
            time.sleep(0.5)                                              # 7, 8, 9 are computed by now
            for k in (0, 1):
                shutil.copy(seq_a / "000002_wd" / "undistorted" / ("0000000%d.png" % k), seq_b / ("%06d_wd" % swapped) / "undistorted" / ("0000000%d.png" % k))
        r = subprocess.run([cli, cfg, str(seq_b / ("%06d_wd" % i))], capture_output=True, text=True, env=env)
        assert r.returncode == 0 and "All done." in r.stdout and r.stdout.isascii(), r.stdout[-1500:]
        time.sleep(0.05)                                                 # a slow caller: the worker gets ahead of it
        src = 2 if i == swapped else i % nd
        for name in NAMES + ("H0_rect.txt", "H1_rect.txt"):
            assert (seq_a / ("%06d_wd" % src) / name).read_bytes() == (seq_b / ("%06d_wd" % i) / name).read_bytes(), f"frame {i}: {name} differs"
    assert _wait_gone(sock)
    rows = {l.split()[0].rstrip("/").rsplit("/", 1)[1]: l.split()[1] for l in tlog.read_text().splitlines() if " total " in l}
    assert rows["%06d_wd" % swapped] == "demand"
    assert sum(1 for v in rows.values() if v == "computed") >= n // 2, rows


def test_speculation_on_a_multi_gpu_node_follows_each_servers_stride(cli, tmp_path):
    """Two pretended GPUs, speculation on: the server of GPU g prepares g + 2, g + 4, ... ahead (never its neighbour's frames), hands
    them to their callers and leaves the workdirs nobody asked for untouched."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("the no-GPU behaviour is tested on the build container")
    seq, cfg = _seq(tmp_path, 12)
    sock = tmp_path / "sock"
    sock.mkdir()
    tlog = tmp_path / "timing.log"
    env = _env(sock, WASS_DEBUG_IMAGES="0", WASS_SERVER_TIMING=str(tlog), WASS_NUM_GPUS="2")
    env.pop("WASS_GPU_DEVICE", None)
    before = sorted(os.listdir(seq / "000011_wd"))
    for i in range(8):
        r = subprocess.run([cli, cfg, str(seq / ("%06d_wd" % i))], capture_output=True, text=True, env=env)
        assert r.returncode == 255 and "no usable MI355X GPU" in r.stdout and ("%06d_wd" % i) in r.stdout
        time.sleep(0.15)
    for i in (8, 9, 10, 11):
        assert sorted(os.listdir(seq / ("%06d_wd" % i))) == before, i      # prepared (and failed on the missing GPU) in memory only
    assert len(_servers(sock)) == 2
    _wait_gone(sock)
    text = tlog.read_text()
    rows = {l.split()[0].rsplit("/", 1)[1]: l.split()[1] for l in text.splitlines() if " total " in l}
    assert [rows["%06d_wd" % i] for i in range(8)] == ["demand"] * 4 + ["computed"] * 4
    assert text.count("speculation: 2 frames computed before they were asked for") == 2 and text.count("dropped unclaimed") == 2
    for i in (0, 4):                                                    # on demand or ahead: the same files
        assert sorted(os.listdir(seq / ("%06d_wd" % i))) == sorted(os.listdir(seq / "000001_wd"))
