"""Seeded synthetic stereo pairs: there is no dataset access, so
bench and tests use band-limited random texture for the left eye and a right
eye resampled with a smooth known disparity field, pushed through the
reference's byte mapping (planar YUV444, ``b ^ 0x80``; preprocess.cpp:975-1040).
"""
from __future__ import annotations

import numpy as np


def _box_blur(a: np.ndarray, r: int) -> np.ndarray:
    if r <= 0:
        return a
    k = 2 * r + 1
    out = a.astype(np.float32)
    for axis in (0, 1):
        pad = [(0, 0)] * out.ndim
        pad[axis] = (r + 1, r)
        c = np.cumsum(np.pad(out, pad, mode="edge"), axis=axis)
        hi = [slice(None)] * out.ndim
        lo = [slice(None)] * out.ndim
        hi[axis] = slice(k, None)
        lo[axis] = slice(0, -k)
        out = (c[tuple(hi)] - c[tuple(lo)]) / k
    return out


def disparity_field(w: int, h: int, dmax: int) -> np.ndarray:
    x = np.arange(w, dtype=np.float32)[None, :]
    y = np.arange(h, dtype=np.float32)[:, None]
    d = dmax / 2.0 * (1.0 + 0.5 * np.sin(2 * np.pi * x / w) * np.cos(2 * np.pi * y / h))
    return np.clip(d, 0, dmax - 1).astype(np.float32)


def stereo_pair_u8(w: int, h: int, dmax: int, seed: int):
    """-> (left, right) uint8 planar 3 x h x w ("Y", "U", "V" planes)."""
    rng = np.random.default_rng(1000 + seed)
    coarse = _box_blur(rng.integers(0, 256, (3, h, w)).astype(np.float32).transpose(1, 2, 0), 6)
    fine = _box_blur(rng.integers(0, 256, (3, h, w)).astype(np.float32).transpose(1, 2, 0), 1)
    left = np.clip(128 + 6.0 * (coarse - 127.5) + 1.2 * (fine - 127.5), 0, 255)
    disp = disparity_field(w, h, dmax)
    xs = np.arange(w, dtype=np.float32)[None, :] - disp      # right(x) = left(x + d)  <=> sample left at x+d
    xs = np.arange(w, dtype=np.float32)[None, :] + disp
    x0 = np.clip(np.floor(xs).astype(np.int64), 0, w - 1)
    x1 = np.clip(x0 + 1, 0, w - 1)
    fx = (xs - np.floor(xs))[..., None]
    rows = np.arange(h)[:, None]
    right = left[rows, x0] * (1 - fx) + left[rows, x1] * fx
    to_u8 = lambda a: np.ascontiguousarray(np.rint(a).astype(np.uint8).transpose(2, 0, 1))
    return to_u8(left), to_u8(right)


def model_input_i8(w: int, h: int, dmax: int, seed: int) -> np.ndarray:
    """int8 NCHW 6 x h x w tensor exactly as CvtNV12Data2Tensors would hand it to Run()."""
    l, r = stereo_pair_u8(w, h, dmax, seed)
    return (np.concatenate([l, r], axis=0) ^ np.uint8(0x80)).view(np.int8)


def random_nv12(w: int, h: int, seed: int) -> np.ndarray:
    """Random w x h NV12 image (h*3/2 rows of w bytes)."""
    rng = np.random.default_rng(seed)
    return rng.integers(0, 256, (h * 3 // 2) * w, dtype=np.uint8)
