#!/usr/bin/env python3
"""Generate the committed golden fixtures under tests/golden/.

Run in the build container only (it imports the reference's Python reader from
/root/reference, which does not exist on the GPU box):

    python tests/golden/make_golden.py

Fixtures
  xyzc_case{0,1}.npz  -- a small organised point cloud, the mesh_cam.xyzC bytes our oracle encoder
                         emits for it, and the points the REFERENCE reader
                         (gridding/wassgridsurface/wass_utils.py:22-35 load_camera_mesh) decodes from
                         those bytes.  Pins the on-disk format.
  rt_from_plane.npz   -- planes and the R,T the REFERENCE computes for them
                         (wass_utils.py:38-48 compute_sea_plane_RT == PovMesh.cpp:1044-1069).
  planes_txt.npz      -- a 3-frame planes.txt (one NaN line) and numpy's nanmean of it
                         (wassgridsurface.py:672-678 semantics for the plane all-reduce).
  sgbm_regress.npz    -- small stereo pairs with the disparity the oracle produces today.  NOT a
                         reference vector (OpenCV is unavailable: parity unpinned); it only guards the
                         oracle and the HIP path against silent drift.
"""
import importlib.util
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True

from oracle import oracle as O  # noqa: E402
from wass_amd import synth  # noqa: E402

spec = importlib.util.spec_from_file_location(
    "ref_wass_utils", "/root/reference/gridding/wassgridsurface/wass_utils.py")
ref = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref)


def cloud(seed, w=37, h=23):
    rng = np.random.default_rng(seed)
    n = np.array([0.05 * (seed + 1), -0.3, 0.95]); n /= np.linalg.norm(n)
    d = -12.5 - seed
    u, v = np.meshgrid(np.arange(w), np.arange(h))
    x = (u - w / 2) * 0.8 + rng.normal(0, 0.05, (h, w))
    y = (v - h / 2) * 0.6 + rng.normal(0, 0.05, (h, w))
    z = (-d - n[0] * x - n[1] * y) / n[2] + rng.normal(0, 0.2, (h, w))
    p3d = np.stack([x, y, z], -1).astype(np.float64)
    valid = (rng.random((h, w)) > 0.3).astype(np.uint8)
    return valid, np.ascontiguousarray(p3d), np.array([*n, d])


for case in range(2):
    valid, p3d, plane = cloud(case)
    blob = O.encode_xyzc(valid, p3d, plane)
    with tempfile.NamedTemporaryFile(suffix=".xyzC", delete=False) as f:
        f.write(blob)
    decoded = ref.load_camera_mesh(f.name)          # 3 x N, float32 math as in the reference
    os.unlink(f.name)
    np.savez_compressed(os.path.join(HERE, f"xyzc_case{case}.npz"), valid=valid, p3d=p3d, plane=plane,
                        xyzc=np.frombuffer(blob, np.uint8), ref_decoded=np.asarray(decoded))

planes = np.array([[0.0123, -0.4567, 0.8895, -11.2], [-0.2, 0.1, 0.9746794344808963, 3.5],
                   [0.3, 0.4, 0.8660254037844386, -0.75], [1e-4, -2e-4, 0.99999997, -20.0]])
planes[:, :3] /= np.linalg.norm(planes[:, :3], axis=1, keepdims=True)    # unit normals
Rs, Ts = [], []
for pl in planes:
    R, T = ref.compute_sea_plane_RT(pl)
    Rs.append(R); Ts.append(np.asarray(T).ravel())
np.savez_compressed(os.path.join(HERE, "rt_from_plane.npz"), planes=planes, R=np.array(Rs), T=np.array(Ts))

frames = ["0.0125 -0.4571 0.8893 -11.25", "nan nan nan nan", "0.0135 -0.4561 0.8898 -11.15"]
arr = np.array([[float(x) for x in l.split()] for l in frames])
np.savez_compressed(os.path.join(HERE, "planes_txt.npz"), text="\n".join(frames) + "\n",
                    planes=arr, nanmean=np.nanmean(arr, axis=0), n_valid=2)

cases = {}
for name, (w, h, D, mode) in {"a": (96, 64, 32, 5), "b": (96, 64, 32, 8), "c": (120, 50, 48, 5)}.items():
    r, l = synth.make_pair(w, h, D, frame_idx=900 + ord(name))
    d, st = O.dense_disparity16(r, l, O.wass_params(D, mode))
    assert not st.overflow
    cases[f"{name}_right"] = r; cases[f"{name}_left"] = l; cases[f"{name}_disp"] = d
    cases[f"{name}_cfg"] = np.array([w, h, D, mode])
np.savez_compressed(os.path.join(HERE, "sgbm_regress.npz"), **cases)
print("golden fixtures written to", HERE)
