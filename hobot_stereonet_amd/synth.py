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


def sbs_nv12_from_planes(left: np.ndarray, right: np.ndarray) -> np.ndarray:
    """Side-by-side NV12 frame (FeedImg's input, stereonet_node.cpp:700-738) from two planar 3 x h x w uint8 eyes:
    2w x h luma rows [left | right] followed by h/2 rows of interleaved UV (chroma = top-left sample of each 2x2)."""
    _, h, w = left.shape
    assert w % 2 == 0 and h % 2 == 0
    out = np.empty((h * 3 // 2, 2 * w), np.uint8)
    for k, eye in enumerate((left, right)):
        out[:h, k * w:(k + 1) * w] = eye[0]
        uv = out[h:, k * w:(k + 1) * w]
        uv[:, 0::2] = eye[1][0::2, 0::2]
        uv[:, 1::2] = eye[2][0::2, 0::2]
    return out.ravel()


def sbs_nv12_frame(w: int, h: int, dmax: int, seed: int) -> np.ndarray:
    """Seeded stereo pair as the side-by-side NV12 frame a camera node would publish."""
    return sbs_nv12_from_planes(*stereo_pair_u8(w, h, dmax, seed))


def sbs_nv12_from_model_input(in6: np.ndarray) -> np.ndarray:
    """The side-by-side NV12 frame whose eyes are the planes of an int8 model tensor (6 x h x w, bytes ^ 0x80)."""
    u8 = np.ascontiguousarray(in6).view(np.uint8) ^ np.uint8(0x80)
    return sbs_nv12_from_planes(u8[:3], u8[3:])
