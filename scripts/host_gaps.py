"""Host-side duration of every call of one bench step (where does the CPU thread wait?)."""
import sys, time
import numpy as np
import torch
sys.path.insert(0, ".")
import wass_amd
from wass_amd import synth

w, h, D = 2456, 2058, 256
dev = torch.device("cuda", 0)
params = wass_amd.default_sgm_params(D, ndirs=8)
ctx = wass_amd.Context(0)
r, l = synth.make_pair(w, h, D, frame_idx=0)
dr, dl = torch.from_numpy(r).to(dev), torch.from_numpy(l).to(dev)
out = torch.empty((h, w), dtype=torch.int16, device=dev)
dispf = torch.empty((h, w), dtype=torch.float32, device=dev)
geom = wass_amd.make_geom(synth.rig_geometry(w, h))
roi = (0, 0, w, h)
burned = (dr <= 254).to(torch.uint8)
host = torch.empty(148 + 6 * w * h, dtype=torch.uint8, pin_memory=True)
acc = {}
def T(name, f):
    t0 = time.perf_counter(); v = f(); acc.setdefault(name, []).append((time.perf_counter() - t0) * 1e6); return v
for i in range(12):
    t_all = time.perf_counter()
    T("sgm_enqueue", lambda: ctx.sgm_disparity_dev(dr, dl, params, out))
    T("post_enqueue", lambda: ctx.disparity_postprocess_dev(out, params, 1, 2, 0, dispf))
    mesh, n = T("triangulate(sync)", lambda: ctx.triangulate_dev(dispf, w, h, roi, roi, geom, dr, None, burned, 20.0, None, 1.0))
    T("remove_outliers(sync)", lambda: mesh.remove_outliers(99.0))
    uv = T("ransac_sample", lambda: wass_amd.ransac_sample(w, h, 400, 12345))
    res = T("fit_plane(sync)", lambda: mesh.fit_plane(uv, 1.0, 1.5))
    pl = np.array(res.plane[:])
    T("encode_async", lambda: mesh.encode_xyzc_async(pl, host.data_ptr(), host.numel()))
    T("mesh.close", lambda: mesh.close())
    T("sgm_timings", lambda: ctx.sgm_timings())
    acc.setdefault("step_total", []).append((time.perf_counter() - t_all) * 1e6)
torch.cuda.synchronize()
for k, v in acc.items():
    print("%-24s %9.1f us" % (k, float(np.mean(v[3:]))))
