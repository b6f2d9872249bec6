import os, sys, time, subprocess, tempfile, shutil
sys.path.insert(0, os.getcwd())
from concurrent.futures import ThreadPoolExecutor
import bench
from wass_amd import build
build.build_host()
tmp = tempfile.mkdtemp(prefix="wass_dbg_", dir="/dev/shm")
if os.environ.get("SPEC_ASAN"):
    bindir = os.path.join(os.getcwd(), "_asan", "bin"); os.makedirs(bindir, exist_ok=True)
    shutil.copy(build.CLI, os.path.join(bindir, "wass_stereo"))
    shutil.copy(os.path.join(os.path.dirname(build.CLI), "asan_wass_stereo_gpu"), os.path.join(bindir, "wass_stereo_gpu"))
    shutil.copy(build.SO, os.path.join(os.getcwd(), "_asan", "libwassgpu.so"))
    build.CLI = os.path.join(bindir, "wass_stereo")
    os.environ["ASAN_OPTIONS"] = "detect_leaks=0:protect_shadow_gap=0:abort_on_error=0"
sock = os.path.join(tmp, "sock"); os.makedirs(sock)
seq, cfg, n = bench.make_sequence(tmp, 8, 6, 8)
tlog = os.path.join(tmp, "t.log")
env = dict(os.environ, WASS_DEBUG_IMAGES="0", WASS_SERVER_DIR=sock, WASS_SERVER_IDLE="4", WASS_SERVER_TIMING=tlog)
env.pop("WASS_NO_SERVER", None)
errf = open(os.path.join(tmp, "server_stderr.txt"), "ab")
env["WASS_SERVER_STDERR"] = "1"
def one(i):
    t = time.perf_counter()
    r = subprocess.run([build.CLI, cfg, os.path.join(seq, "%06d_wd" % i)], stdout=subprocess.PIPE, stderr=errf, env=env)
    return i, r.returncode, time.perf_counter() - t
one(0)
for par in (8, 8):
    with ThreadPoolExecutor(par) as ex:
        t1 = time.perf_counter(); res = list(ex.map(one, range(1, n))); t2 = time.perf_counter()
    time.sleep(0.7); print(par, "callers:", round((n - 1) / (t2 - t1), 1), "pairs/s; slowest", sorted((round(s, 3), i) for i, rc, s in res)[-6:], "failed", sum(1 for _, rc, _ in res if rc))
time.sleep(5)
raw = open(tlog, "rb").read()
for l in raw.decode("latin1").splitlines():
    w = l.split()
    if " total " in l and float(w[w.index("total") + 1]) > 70: print(l[-150:])
    if "speculation" in l or "read-ahead" in l: print(l)
print("SERVER STDERR:", open(os.path.join(tmp, "server_stderr.txt"), "rb").read()[:12000].decode("latin1"))
shutil.rmtree(tmp, ignore_errors=True)
