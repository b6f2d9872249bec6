import sys, os, glob, re, subprocess, tempfile, shutil, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
from wass_amd import build
build.build_host()
tmp = tempfile.mkdtemp(prefix="wass_lat_", dir="/dev/shm")
seq, cfg, n = bench.make_sequence(tmp, 8, 12, 8)
for env in ("WASS_HOST_INLIER_TEXT=0", "WASS_HOST_INLIER_TEXT=1", "WASS_HOST_INLIER_TEXT=0", "WASS_HOST_INLIER_TEXT=1"):
    for wd in glob.glob(os.path.join(seq, "*_wd")):
        for f in ("mesh_cam.xyzC", "plane.txt", "plane_refinement_inliers.xyz", "wass_stereo_log.txt"):
            try: os.remove(os.path.join(wd, f))
            except OSError: pass
    e = dict(os.environ, WASS_PIPE_TIMING="1"); k, v = env.split("="); e[k] = v
    r = subprocess.run([build.BATCH, cfg, "--sequence", seq, "--gpus", "1"], capture_output=True, text=True, env=e)
    steady = [l for l in r.stdout.splitlines() if l.startswith("steady")]
    sub, out, rect = [], [], []
    for wd in sorted(glob.glob(os.path.join(seq, "*_wd")))[16:]:
        log = open(os.path.join(wd, "wass_stereo_log.txt")).read()
        m = re.search(r"submission to result ([0-9.e-]+) s, output ([0-9.e-]+) s", log)
        if m: sub.append(float(m.group(1))); out.append(float(m.group(2)))
        m = re.search(r"Rectification\s+\|\s+([0-9.e-]+)", log)
        if m: rect.append(float(m.group(1)))
    print([l for l in r.stderr.splitlines() if "submit() per frame" in l])
    import statistics as st
    print(env, steady[0].split(":")[1].strip() if steady else None, "| submission->result median %.1f ms, output %.1f ms, 'Rectification' row (plan + submit host time) %.1f ms" % (1e3*st.median(sub), 1e3*st.median(out), 1e3*st.median(rect)), flush=True)
shutil.rmtree(tmp, ignore_errors=True)
