"""Deterministic synthetic rectified stereo pairs.

A textured, tilted "sea plane": the left image is an analytic sum of 24
sinusoids, the right image is the same texture resampled at ``x - d*(x, y)``
where ``d*`` is affine in (x, y) (an exact 3-D plane) plus a +-0.6 px ripple.
Orientation matches the reference call ``compute(right_image, left_image)``
(src/wass_stereo/wass_stereo.cpp:837): right pixel x matches left column
x - d, so the disparity map lives in the right image's frame
(``xl = xr - d``, wass_stereo.cpp:1180).

Pure numpy; used by tests, bench.py and smoke() to build inputs. It is data
generation only -- no part of the stereo algorithm lives here.
"""
from __future__ import annotations

import numpy as np

SEED_BASE = 0x5741535300000000
_M64 = (1 << 64) - 1


def _splitmix64(state: int):
    """One splitmix64 step -> (new_state, output)."""
    state = (state + 0x9E3779B97F4A7C15) & _M64
    z = state
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _M64
    return state, z ^ (z >> 31)


def _uniform(state: int):
    state, z = _splitmix64(state)
    return state, (z >> 11) / float(1 << 53)


def _hash_noise(seed: int, cam: int, h: int, w: int) -> np.ndarray:
    """Per-pixel integer noise in [-3, 3] from a 64-bit mix of (seed, cam, y, x)."""
    y = np.arange(h, dtype=np.uint64)[:, None]
    x = np.arange(w, dtype=np.uint64)[None, :]
    with np.errstate(over="ignore"):
        z = (np.uint64(seed & _M64) + np.uint64(0x9E3779B97F4A7C15) * (y * np.uint64(65536) + x + np.uint64(1))
             + np.uint64(0xD1B54A32D192ED03) * np.uint64(cam + 1))
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return (z % np.uint64(7)).astype(np.int64) - 3


def texture_params(frame_idx: int = 0, n_waves: int = 24):
    state = (SEED_BASE + frame_idx) & _M64
    A, f, g, phi = [], [], [], []
    lo, hi = 1.0 / 256.0, 1.0 / 6.0
    for _ in range(n_waves):
        state, r = _uniform(state); A.append(2.0 + 8.0 * r)
        state, r = _uniform(state); f.append(lo + (hi - lo) * r)
        state, r = _uniform(state); gg = lo + (hi - lo) * r
        state, r = _uniform(state); g.append(gg if r < 0.5 else -gg)
        state, r = _uniform(state); phi.append(2.0 * np.pi * r)
    return np.array(A), np.array(f), np.array(g), np.array(phi)


def true_disparity(w: int, h: int, num_disp: int) -> np.ndarray:
    """d*(x, y) in the right image's frame, float64 (h, w)."""
    x = np.arange(w, dtype=np.float64)[None, :]
    y = np.arange(h, dtype=np.float64)[:, None]
    d0, s = 0.15 * num_disp, 0.6 * num_disp
    return d0 + s * (y / h) + 0.4 * np.sin(2 * np.pi * x / 97.0) + 0.2 * np.sin(2 * np.pi * y / 61.0)


def _tex(u: np.ndarray, v: np.ndarray, params) -> np.ndarray:
    A, f, g, phi = params
    out = np.full(np.broadcast(u, v).shape, 128.0)
    for k in range(len(A)):
        out += A[k] * np.sin(2 * np.pi * (f[k] * u + g[k] * v) + phi[k])
    return out


def make_pair(w: int, h: int, num_disp: int, frame_idx: int = 0, noise: bool = True):
    """Return (right, left) u8 (h, w) C-contiguous arrays."""
    params = texture_params(frame_idx)
    x = np.arange(w, dtype=np.float64)[None, :]
    y = np.arange(h, dtype=np.float64)[:, None]
    left = _tex(x, y, params)
    right = _tex(x - true_disparity(w, h, num_disp), y, params)
    if noise:
        seed = (SEED_BASE + frame_idx) & _M64
        left = left + _hash_noise(seed, 0, h, w)
        right = right + _hash_noise(seed, 1, h, w)
    left = np.clip(np.rint(left), 1, 254).astype(np.uint8)
    right = np.clip(np.rint(right), 1, 254).astype(np.uint8)
    return np.ascontiguousarray(right), np.ascontiguousarray(left)


def _i64(c: int) -> int:
    """64-bit constant as the int64 with the same bit pattern."""
    c &= _M64
    return c - (1 << 64) if c >= (1 << 63) else c


def make_pair_torch(w: int, h: int, num_disp: int, frame_idx: int = 0, device="cpu"):
    """make_pair() evaluated with torch (float64 sines, the 64-bit hash in wrapping int64 arithmetic) on `device`:
    the same images (up to the last bit of a sine that falls exactly on a rounding boundary, which has not been
    observed), in milliseconds on a GPU instead of seconds -- bench.py uses it to prepare 64 distinct full-size frames.
    Returns (right, left) uint8 tensors on `device`."""
    import torch
    A, f, g, phi = texture_params(frame_idx)
    x = torch.arange(w, dtype=torch.float64, device=device)[None, :]
    y = torch.arange(h, dtype=torch.float64, device=device)[:, None]
    d0, s = 0.15 * num_disp, 0.6 * num_disp
    disp = d0 + s * (y / h) + 0.4 * torch.sin(2 * np.pi * x / 97.0) + 0.2 * torch.sin(2 * np.pi * y / 61.0)

    def tex(u, v):
        out = torch.full((h, w), 128.0, dtype=torch.float64, device=device)
        for k in range(len(A)):
            out += float(A[k]) * torch.sin(2 * np.pi * (float(f[k]) * u + float(g[k]) * v) + float(phi[k]))
        return out

    def noise(cam):
        seed = (SEED_BASE + frame_idx) & _M64
        yy = torch.arange(h, dtype=torch.int64, device=device)[:, None]
        xx = torch.arange(w, dtype=torch.int64, device=device)[None, :]
        z = _i64(seed) + _i64(0x9E3779B97F4A7C15) * (yy * 65536 + xx + 1) + _i64(0xD1B54A32D192ED03 * (cam + 1))
        for sh, mul in ((30, 0xBF58476D1CE4E5B9), (27, 0x94D049BB133111EB)):
            z = (z ^ ((z >> sh) & ((1 << (64 - sh)) - 1))) * _i64(mul)        # logical shift, wrapping multiply
        z = z ^ ((z >> 31) & ((1 << 33) - 1))
        # unsigned remainder of the bit pattern: U = z + 2^64 for z < 0, and 2^64 = 2 (mod 7)
        return torch.remainder(torch.remainder(z, 7) + 2 * (z < 0).to(torch.int64), 7) - 3

    left = tex(x, y) + noise(0)
    right = tex(x - disp, y) + noise(1)
    to_u8 = lambda t: torch.clamp(torch.round(t), 1, 254).to(torch.uint8).contiguous()
    return to_u8(right), to_u8(left)


def rig_geometry(w: int, h: int):
    """Ideal rig of SURVEY.md 8(d): K0=K1, R=I, T=(1,0,0), identity rectification."""
    f = 0.9 * w
    K = np.array([[f, 0, w / 2.0], [0, f, h / 2.0], [0, 0, 1.0]])
    P = np.hstack([K, np.zeros((3, 1))])
    return dict(K_left=K.copy(), K_right=K.copy(), R=np.eye(3), T=np.array([1.0, 0.0, 0.0]),
                R1=np.eye(3), R2=np.eye(3), P1=P.copy(), P2=P.copy(),
                HLi=np.eye(3), HRi=np.eye(3))
