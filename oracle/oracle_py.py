"""ctypes binding of the CPU oracle — TEST INFRASTRUCTURE ONLY.

Importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
(as the checker / reported baseline), never from hobot_stereonet_amd/.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_DIR = os.path.dirname(os.path.abspath(__file__))
_SAN = os.environ.get("SN_SANITIZE") == "1"      # scripts/run_sanitized.sh: the ASan + UBSan build of the checker
_LIB = os.path.join(_DIR, "libstereonet_oracle_asan.so" if _SAN else "libstereonet_oracle.so")
_lib = None


def build(force: bool = False) -> str:
    src = [os.path.join(_DIR, f) for f in ("stereonet_oracle.c", "stereonet_oracle.h")]
    stale = (not os.path.exists(_LIB)) or any(os.path.getmtime(s) > os.path.getmtime(_LIB) for s in src)
    if force or stale:
        subprocess.check_call(["make", "-C", _DIR, "-s", "-B", os.path.basename(_LIB)])
    return _LIB


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            build()
        _lib = C.CDLL(_LIB)
        _lib.so_weight_count.restype = C.c_long
        _lib.so_weight_offset.restype = C.c_long
        _lib.so_weight_offset.argtypes = [C.c_char_p]
        _lib.so_quantize.restype = C.c_int8
        _lib.so_quantize.argtypes = [C.c_float] * 5
        _lib.so_forward.restype = C.c_int
        _lib.so_forward_levels.restype = C.c_int
        _lib.so_weight_count_levels.restype = C.c_long
        _lib.so_weight_count_levels.argtypes = [C.c_int]
    return _lib


def _p(a, t=C.c_void_p):
    return a.ctypes.data_as(t) if a is not None else None


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def num_threads() -> int:
    return lib().so_num_threads()


def weight_count() -> int:
    return lib().so_weight_count()


MULTI_LEVELS = 4


def weight_count_levels(levels: int) -> int:
    return lib().so_weight_count_levels(levels)


def levels_of(weights: np.ndarray) -> int:
    for lv in (1, MULTI_LEVELS):
        if weights.size == weight_count_levels(lv):
            return lv
    raise ValueError(f"{weights.size} parameters: neither a single nor a multi blob")


def weight_offset(name: str) -> int:
    return lib().so_weight_offset(name.encode())


def split_sbs_nv12(sbs: np.ndarray, w: int, h: int):
    sbs = np.ascontiguousarray(sbs, dtype=np.uint8)
    n = w * h * 3 // 2
    left = np.empty(n, np.uint8)
    right = np.empty(n, np.uint8)
    lib().so_split_sbs_nv12(_p(sbs), C.c_int(w), C.c_int(h), _p(left), _p(right))
    return left, right


def yuv420_to_yuv444(img: np.ndarray, w: int, h: int) -> np.ndarray:
    img = np.ascontiguousarray(img, dtype=np.uint8)
    out = np.empty(3 * w * h, np.uint8)
    lib().so_yuv420_to_yuv444(_p(img), _p(out), C.c_int(w), C.c_int(h))
    return out.reshape(3, h, w)


def quantize(v: float) -> int:
    return int(lib().so_quantize(v, 0.0078125, 0.5, -128.0, 127.0))


def preprocess_nv12(left: np.ndarray, right: np.ndarray, w: int, h: int) -> np.ndarray:
    left = np.ascontiguousarray(left, dtype=np.uint8)
    right = np.ascontiguousarray(right, dtype=np.uint8)
    out = np.empty(6 * w * h, np.int8)
    lib().so_preprocess_nv12(_p(left), _p(right), C.c_int(w), C.c_int(h), _p(out))
    return out.reshape(6, h, w)


def bgr_to_nv12(bgr: np.ndarray) -> np.ndarray:
    """(h, w, 3) uint8 B,G,R -> flat NV12 (w*h*3/2 bytes)."""
    bgr = np.ascontiguousarray(bgr, dtype=np.uint8)
    h, w = bgr.shape[:2]
    out = np.empty(w * h * 3 // 2, np.uint8)
    if lib().so_bgr_to_nv12(_p(bgr), C.c_int(w), C.c_int(h), _p(out)) != 0:
        raise ValueError("width and height must be even")
    return out


def dequant_depth(raw: np.ndarray, scale: float, dmax: float):
    raw = np.ascontiguousarray(raw, dtype=np.int32)
    disp = np.empty(raw.shape, np.float32)
    depth = np.empty(raw.shape, np.float32)
    lib().so_dequant_depth(_p(raw), C.c_int(raw.size), C.c_float(scale), C.c_float(dmax), _p(disp), _p(depth))
    return disp, depth


def conv2d(x, wt, bias, stride=1, pad=0, dil=1):
    x, wt = _f32(x), _f32(wt)
    cin, h, w = x.shape
    cout, _, k, _ = wt.shape
    ho = (h + 2 * pad - dil * (k - 1) - 1) // stride + 1
    wo = (w + 2 * pad - dil * (k - 1) - 1) // stride + 1
    out = np.empty((cout, ho, wo), np.float32)
    b = _f32(bias) if bias is not None else None
    lib().so_conv2d(_p(x), C.c_int(cin), C.c_int(h), C.c_int(w), _p(wt), _p(b), C.c_int(cout),
                    C.c_int(k), C.c_int(stride), C.c_int(pad), C.c_int(dil), _p(out))
    return out


def conv3d(x, wt, bias):
    x, wt = _f32(x), _f32(wt)
    cin, d, h, w = x.shape
    cout = wt.shape[0]
    out = np.empty((cout, d, h, w), np.float32)
    b = _f32(bias) if bias is not None else None
    lib().so_conv3d(_p(x), C.c_int(cin), C.c_int(d), C.c_int(h), C.c_int(w), _p(wt), _p(b), C.c_int(cout), _p(out))
    return out


def cost_volume(fl, fr, dl):
    fl, fr = _f32(fl), _f32(fr)
    c, h, w = fl.shape
    out = np.empty((c, dl, h, w), np.float32)
    lib().so_cost_volume(_p(fl), _p(fr), C.c_int(c), C.c_int(dl), C.c_int(h), C.c_int(w), _p(out))
    return out


def soft_argmin(cost):
    cost = _f32(cost)
    dl, h, w = cost.shape
    out = np.empty((h, w), np.float32)
    lib().so_soft_argmin(_p(cost), C.c_int(dl), C.c_int(h), C.c_int(w), _p(out))
    return out


def upsample_bilinear(x, factor, mul=1.0):
    x = _f32(x)
    h, w = x.shape
    out = np.empty((h * factor, w * factor), np.float32)
    lib().so_upsample_bilinear(_p(x), C.c_int(h), C.c_int(w), C.c_int(factor), C.c_float(mul), _p(out))
    return out


def features(weights, planes):
    weights, planes = _f32(weights), _f32(planes)
    _, hp, wp = planes.shape
    out = np.empty((32, hp // 16, wp // 16), np.float32)
    lib().so_features(_p(weights), _p(planes), C.c_int(hp), C.c_int(wp), _p(out))
    return out


def aggregate(weights, fl, fr, dl):
    weights, fl, fr = _f32(weights), _f32(fl), _f32(fr)
    _, hl, wl = fl.shape
    out = np.empty((dl, hl, wl), np.float32)
    lib().so_aggregate(_p(weights), _p(fl), _p(fr), C.c_int(dl), C.c_int(hl), C.c_int(wl), _p(out))
    return out


def refine(weights, disp_up, img, dmax):
    weights, disp_up, img = _f32(weights), _f32(disp_up), _f32(img)
    hp, wp = disp_up.shape
    out = np.empty((hp, wp), np.float32)
    lib().so_refine(_p(weights), _p(disp_up), _p(img), C.c_int(hp), C.c_int(wp), C.c_int(dmax), _p(out))
    return out


def avgpool2(x):
    x = _f32(x)
    c, h, w = x.shape
    out = np.empty((c, h // 2, w // 2), np.float32)
    lib().so_avgpool2(_p(x), C.c_int(c), C.c_int(h), C.c_int(w), _p(out))
    return out


def refine_level(weights, level, disp_up, img, dnorm):
    weights, disp_up, img = _f32(weights), _f32(disp_up), _f32(img)
    hp, wp = disp_up.shape
    out = np.empty((hp, wp), np.float32)
    lib().so_refine_level(_p(weights), C.c_int(level), _p(disp_up), _p(img), C.c_int(hp), C.c_int(wp),
                          C.c_float(dnorm), _p(out))
    return out


def forward_levels(weights, in6, dmax):
    """in6: int8 (6,h,w) -> (disp f32 (h,w), raw int32 (h,w), disp_low f32, [level-1 map, level-2 map, ...]).
    The number of refinement levels follows from the size of the weight blob (single: no level maps)."""
    weights = _f32(weights)
    levels = levels_of(weights)
    in6 = np.ascontiguousarray(in6, dtype=np.int8)
    _, h, w = in6.shape
    hp, wp = (h + 15) // 16 * 16, (w + 15) // 16 * 16
    disp = np.empty((h, w), np.float32)
    raw = np.empty((h, w), np.int32)
    low = np.empty((hp // 16, wp // 16), np.float32)
    maps = [np.empty((hp >> k, wp >> k), np.float32) for k in range(1, levels)]
    ptrs = (C.c_void_p * max(1, len(maps)))(*[m.ctypes.data for m in maps]) if maps else None
    rc = lib().so_forward_levels(_p(weights), C.c_int(levels), _p(in6), C.c_int(w), C.c_int(h), C.c_int(dmax),
                                 _p(disp), _p(raw), _p(low), ptrs)
    if rc != 0:
        raise ValueError("so_forward_levels rejected its arguments")
    return disp, raw, low, maps


def forward(weights, in6, dmax):
    """in6: int8 (6,h,w) -> (disp f32 (h,w), raw int32 (h,w), disp_low f32); single or multi by the blob's size"""
    disp, raw, low, _ = forward_levels(weights, in6, dmax)
    return disp, raw, low
