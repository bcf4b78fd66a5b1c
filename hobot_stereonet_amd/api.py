"""ctypes binding of libstereonet_hip.so (include/stereonet_hip.h) — the Python face of the
drop-in boundary.  It mirrors the call sequence of the reference node:

    Init()/GetModelInputSize  -> StereoNetHIP(model_file)          stereonet_node.cpp:44-45
    Run(inputs, out, sync)    -> .infer(...) / .submit()+.wait()    stereonet_node.cpp:812,968
    CvtNV12Data2Tensors       -> .preprocess_nv12(...)              preprocess.cpp:913-1059

There is no CPU fallback: constructing StereoNetHIP without a gfx950 device raises.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import numpy as np

from . import build as _build

SN_MEM_HOST, SN_MEM_DEVICE = 0, 1
PREC_DEFAULT, PREC_F16X3, PREC_F16, PREC_FP32, PREC_AUTO = 0, 1, 2, 3, 4      # include/stereonet_hip.h; 0 selects PREC_AUTO
PREC_NAMES = {PREC_F16X3: "f16x3", PREC_F16: "f16", PREC_FP32: "fp32", PREC_AUTO: "auto"}
ABI_VERSION = 3
STAGES = ("features", "aggregate", "refine", "refine_conv", "total", "dominant")


class SnConfig(C.Structure):
    _fields_ = [("device", C.c_int), ("max_batch", C.c_int), ("width", C.c_int), ("height", C.c_int),
                ("dmax", C.c_int), ("precision", C.c_int), ("task_num", C.c_int), ("refine_chunk", C.c_int),
                ("piece", C.c_int)]


class SnIoInfo(C.Structure):
    _fields_ = [("width", C.c_int), ("height", C.c_int), ("dmax", C.c_int), ("in_channels", C.c_int),
                ("max_batch", C.c_int), ("precision", C.c_int), ("task_num", C.c_int), ("device", C.c_int),
                ("out_scale", C.c_float), ("in_bytes", C.c_size_t), ("out_bytes", C.c_size_t),
                ("flops_per_pair", C.c_double), ("refine_chunk", C.c_int), ("piece", C.c_int),
                ("tower_streams", C.c_int), ("refine_levels", C.c_int), ("precision_selected", C.c_int)]


class SnRefineStats(C.Structure):
    """sn_refine_stats (include/stereonet_hip.h): the refinement statistic and SN_PREC_AUTO's state."""
    _fields_ = [("levels", C.c_int), ("precision", C.c_int), ("precision_selected", C.c_int), ("precision_last", C.c_int),
                ("calls", C.c_uint64), ("pairs", C.c_uint64), ("switches", C.c_uint64), ("reruns", C.c_uint64),
                ("level_px", C.c_double * 4), ("residual_px", C.c_double), ("running_px", C.c_double),
                ("envelope_px", C.c_double), ("limit_px", C.c_double), ("selfcheck_epe_px", C.c_double),
                ("selfcheck_residual_px", C.c_double)]


class SnAutoState(C.Structure):
    """sn_auto_state: SN_PREC_AUTO's state machine (pure functions sn_auto_*)."""
    _fields_ = [("mode", C.c_int), ("calm", C.c_int), ("envelope_px", C.c_double), ("epe_per_px", C.c_double),
                ("running_px", C.c_double), ("switches", C.c_uint64)]


class StereoNetError(RuntimeError):
    def __init__(self, code: int, where: str, detail: str = ""):
        self.code = code
        super().__init__(f"{where}: {error_string(code)} (code {code}){': ' + detail if detail else ''}")


_lib = None


def load_library(path: Optional[str] = None):
    """Loads (building if stale and hipcc is present) libstereonet_hip.so."""
    global _lib
    if _lib is not None:
        return _lib
    path = path or os.environ.get("STEREONET_HIP_LIB") or _build.LIB      # STEREONET_HIP_LIB: A/B builds of the library
    # One HIP runtime per process: the PyTorch wheel bundles its own libamdhip64/libhsa-runtime64.  If this library
    # came in first it would bind the system ROCm copies, torch would later add its own, and whichever runtime
    # opened the device second would report "no ROCm-capable device".  Importing torch first makes both share
    # torch's copy (same SONAME); without torch installed the system runtime is the only one.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    if not os.path.exists(path) or (os.path.exists(_build.HIPCC) and _build.is_stale()):
        _build.build()
    lib = C.CDLL(path)
    vp, ip, i8p, i32p, fp, u8p = C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p
    lib.sn_create.argtypes = [C.c_char_p, C.POINTER(SnConfig), C.POINTER(vp)]
    lib.sn_destroy.argtypes = [vp]
    lib.sn_get_io_info.argtypes = [vp, C.POINTER(SnIoInfo)]
    lib.sn_strerror.restype = C.c_char_p
    lib.sn_strerror.argtypes = [ip]
    lib.sn_last_error.restype = C.c_char_p
    lib.sn_last_error.argtypes = [vp]
    lib.sn_infer_i8.argtypes = [vp, i8p, i32p, fp, ip, vp]
    lib.sn_infer_batch.argtypes = [vp, ip, i8p, i32p, fp, ip, vp]
    lib.sn_preprocess_nv12.argtypes = [vp, u8p, u8p, ip, ip, i8p, ip, vp]
    lib.sn_infer_sbs_nv12.argtypes = [vp, u8p, ip, ip, i32p, fp, i8p, ip, vp]
    lib.sn_preprocess_sbs_nv12_batch.argtypes = [vp, ip, u8p, ip, ip, i8p, ip, vp]
    lib.sn_submit.argtypes = [vp, i8p, i32p, fp, ip, C.POINTER(C.c_uint64)]
    lib.sn_submit_nv12.argtypes = [vp, u8p, ip, ip, i32p, fp, ip, C.POINTER(C.c_uint64)]
    lib.sn_wait.argtypes = [vp, C.c_uint64, C.POINTER(C.c_float)]
    lib.sn_synchronize.argtypes = [vp]
    lib.sn_set_profiling.argtypes = [vp, ip]
    lib.sn_get_stage_ms.argtypes = [vp, C.POINTER(C.c_float), ip]
    lib.sn_get_dominant_kernel.argtypes = [vp, C.c_char_p, C.c_size_t, C.POINTER(ip), C.POINTER(C.c_double),
                                           C.POINTER(C.c_double)]
    lib.sn_mgpu_shard.argtypes = [ip, ip, ip, C.POINTER(ip), C.POINTER(ip)]
    lib.sn_mgpu_create.argtypes = [C.c_char_p, C.POINTER(SnConfig), C.POINTER(ip), ip, C.POINTER(vp)]
    lib.sn_mgpu_destroy.argtypes = [vp]
    lib.sn_mgpu_get_info.argtypes = [vp, C.POINTER(ip), C.POINTER(ip), C.POINTER(ip)]
    lib.sn_mgpu_get_handle.argtypes = [vp, ip, C.POINTER(vp)]
    lib.sn_mgpu_infer_batch.argtypes = [vp, ip, i8p, i32p, fp]
    lib.sn_mgpu_infer_batch_device.argtypes = [vp, ip, C.POINTER(vp), i32p, fp]
    lib.sn_mgpu_submit_device.argtypes = [vp, ip, C.POINTER(vp), i32p, fp, C.POINTER(C.c_uint64)]
    lib.sn_mgpu_wait.argtypes = [vp, C.c_uint64]
    lib.sn_mgpu_ring_init.argtypes = [vp]
    lib.sn_mgpu_ring_submit.argtypes = [vp, C.POINTER(C.c_uint64), C.POINTER(ip)]
    lib.sn_mgpu_ring_wait.argtypes = [vp, C.c_uint64, C.POINTER(ip)]
    lib.sn_mgpu_last_error.restype = C.c_char_p
    lib.sn_mgpu_last_error.argtypes = [vp]
    lib.sn_dbg_conv2d.argtypes = [vp, fp, ip, ip, ip, fp, fp, ip, ip, ip, ip, fp, fp]
    lib.sn_dbg_down0.argtypes = [vp, i8p, ip, ip, fp, fp, ip, fp]
    lib.sn_dbg_compose_down01.argtypes = [fp, fp, fp, fp, fp, fp]
    lib.sn_dbg_round_kernels_f16.argtypes = [fp, ip, fp]
    lib.sn_dbg_down01.argtypes = [vp, i8p, ip, ip, fp, fp, fp, fp, fp]
    lib.sn_dbg_refin.argtypes = [vp, fp, i8p, ip, ip, ip, fp, fp, ip, fp]
    lib.sn_dbg_conv3d.argtypes = [vp, fp, ip, ip, ip, fp, fp, ip, fp]
    lib.sn_dbg_ref_conv_f16.argtypes = [vp, fp, ip, ip, fp, fp, ip, ip, fp, fp]
    lib.sn_dbg_ref_conv_f16x3.argtypes = [vp, fp, ip, ip, fp, fp, ip, ip, fp, fp]
    lib.sn_dbg_ref_block_f16.argtypes = [vp, fp, ip, ip, fp, fp, fp, fp, ip, fp]
    lib.sn_dbg_ref_block_f16x3.argtypes = [vp, fp, ip, ip, fp, fp, fp, fp, ip, ip, fp]
    lib.sn_dbg_ref_tail_f16.argtypes = [vp, ip, fp, ip, ip, fp, fp, fp, fp, fp, C.c_float, fp, ip, C.c_float, ip, ip, ip, fp, i32p]
    lib.sn_dbg_read.argtypes = [vp, C.c_char_p, fp, C.c_size_t, C.POINTER(C.c_size_t)]
    lib.sn_dbg_copy_limited.argtypes = [vp, vp, C.c_size_t, ip, vp]
    lib.sn_depth_from_raw.argtypes = [vp, ip, i32p, C.c_float, C.c_float, fp, fp, ip, vp]
    lib.sn_get_refine_stats.argtypes = [vp, C.POINTER(SnRefineStats)]
    lib.sn_auto_init.argtypes = [C.POINTER(SnAutoState), ip]
    lib.sn_auto_observe.argtypes = [C.POINTER(SnAutoState), C.c_double]
    lib.sn_auto_limit_px.argtypes = [C.POINTER(SnAutoState)]
    lib.sn_auto_limit_px.restype = C.c_double
    lib.sn_auto_envelope_px.argtypes = [ip]
    lib.sn_auto_envelope_px.restype = C.c_double
    for name in ("sn_create", "sn_destroy", "sn_get_io_info", "sn_infer_i8", "sn_infer_batch", "sn_preprocess_nv12",
                 "sn_infer_sbs_nv12", "sn_preprocess_sbs_nv12_batch", "sn_submit", "sn_submit_nv12", "sn_wait", "sn_synchronize", "sn_set_profiling",
                 "sn_get_stage_ms", "sn_get_dominant_kernel", "sn_mgpu_shard", "sn_mgpu_create", "sn_mgpu_destroy",
                 "sn_mgpu_get_info", "sn_mgpu_get_handle", "sn_mgpu_infer_batch", "sn_mgpu_infer_batch_device",
                 "sn_mgpu_submit_device", "sn_mgpu_wait", "sn_mgpu_ring_init", "sn_mgpu_ring_submit", "sn_mgpu_ring_wait", "sn_dbg_conv2d", "sn_dbg_down0", "sn_dbg_compose_down01", "sn_dbg_round_kernels_f16", "sn_dbg_down01", "sn_dbg_refin", "sn_dbg_conv3d", "sn_dbg_ref_conv_f16", "sn_dbg_ref_conv_f16x3", "sn_dbg_ref_block_f16", "sn_dbg_ref_block_f16x3", "sn_dbg_ref_tail_f16", "sn_dbg_read", "sn_dbg_copy_limited", "sn_depth_from_raw", "sn_get_refine_stats", "sn_auto_init", "sn_auto_observe"):
        getattr(lib, name).restype = C.c_int
    _lib = lib
    return lib


def round_kernels_f16(w):
    """The fp16 rounding SN_PREC_F16 applies to its tower's 3x3 weights at model load (host only): w (..., 3, 3) float32 ->
    same shape, fp16-representable float32, the sum of every kernel's nine rounding errors minimised."""
    a = np.ascontiguousarray(w, np.float32)
    assert a.shape[-2:] == (3, 3)
    out = np.empty_like(a)
    rc = load_library().sn_dbg_round_kernels_f16(a.ctypes.data, a.size // 9, out.ctypes.data)
    if rc:
        raise StereoNetError(rc, "sn_dbg_round_kernels_f16")
    return out


def compose_down01(w0, b0, w1, b1):
    """Host-side fold of the first two down-convs (no device needed): -> (weff (9,32,3,13,13), beff (9,32)) float32;
    class = 3 * row class + column class, each {first, inner, last} row / column of the quarter-resolution map."""
    w0, b0, w1, b1 = (np.ascontiguousarray(a, np.float32) for a in (w0, b0, w1, b1))
    assert w0.shape == (32, 3, 5, 5) and w1.shape == (32, 32, 5, 5) and b0.shape == (32,) and b1.shape == (32,)
    weff = np.empty((9, 32, 3, 13, 13), np.float32)
    beff = np.empty((9, 32), np.float32)
    rc = load_library().sn_dbg_compose_down01(w0.ctypes.data, b0.ctypes.data, w1.ctypes.data, b1.ctypes.data,
                                              weff.ctypes.data, beff.ctypes.data)
    if rc:
        raise StereoNetError(rc, "sn_dbg_compose_down01")
    return weff, beff


def error_string(code: int) -> str:
    return load_library().sn_strerror(code).decode()


def _np_ptr(a: Optional[np.ndarray]):
    return a.ctypes.data if a is not None else None


class StereoNetHIP:
    """One GPU's StereoNet engine (sn_handle)."""

    def __init__(self, model_file: str, device: int = -1, max_batch: int = 1, width: int = 0, height: int = 0,
                 dmax: int = 0, precision: int = PREC_DEFAULT, task_num: int = 4, refine_chunk: int = 0,
                 piece: int = 0):
        self._lib = load_library()
        self._h = C.c_void_p()
        cfg = SnConfig(device, max_batch, width, height, dmax, precision, task_num, refine_chunk, piece)
        rc = self._lib.sn_create(model_file.encode(), C.byref(cfg), C.byref(self._h))
        if rc != 0:
            self._h = C.c_void_p()
            detail = self._lib.sn_last_error(None)
            raise StereoNetError(rc, f"sn_create({model_file!r})", detail.decode() if detail else "")
        info = SnIoInfo()
        self._check(self._lib.sn_get_io_info(self._h, C.byref(info)), "sn_get_io_info")
        self.info = info
        self.width, self.height, self.dmax = info.width, info.height, info.dmax
        self.max_batch = info.max_batch
        self.refine_chunk, self.piece, self.tower_streams = info.refine_chunk, info.piece, info.tower_streams
        self.refine_levels = info.refine_levels
        self.precision = info.precision
        self.out_scale = float(info.out_scale)
        self.flops_per_pair = float(info.flops_per_pair)

    # -- plumbing -----------------------------------------------------------------------------
    def _check(self, rc: int, where: str):
        if rc != 0:
            detail = self._lib.sn_last_error(self._h).decode() if self._h else ""
            raise StereoNetError(rc, where, detail)

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self._lib.sn_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # -- refinement statistic / SN_PREC_AUTO ----------------------------------------------------
    def refine_stats(self) -> dict:
        """sn_get_refine_stats as a dict (precisions as names): what the refinement moved the last call's maps by, and
        which arithmetic an SN_PREC_AUTO handle is in."""
        st = SnRefineStats()
        self._check(self._lib.sn_get_refine_stats(self._h, C.byref(st)), "sn_get_refine_stats")
        d = {k: getattr(st, k) for k, _ in SnRefineStats._fields_ if k != "level_px"}
        d["level_px"] = [float(v) for v in st.level_px][:max(1, st.levels)]
        for k in ("precision", "precision_selected", "precision_last"):
            d[k] = PREC_NAMES.get(d[k], str(d[k]))
        return d

    @property
    def precision_selected(self) -> int:
        info = SnIoInfo()
        self._check(self._lib.sn_get_io_info(self._h, C.byref(info)), "sn_get_io_info")
        return info.precision_selected

    # -- Run (host numpy buffers) ---------------------------------------------------------------
    def infer(self, in6: np.ndarray, want_disp: bool = True, want_raw: bool = True):
        """in6: int8 (6,H,W) or (n,6,H,W) -> (disp float32, raw int32) with matching leading dims."""
        x = np.ascontiguousarray(in6, dtype=np.int8)
        single = x.ndim == 3
        if single:
            x = x[None]
        n = x.shape[0]
        if x.shape[1:] != (6, self.height, self.width):
            raise StereoNetError(-1, "infer", f"input shape {x.shape} != (n,6,{self.height},{self.width})")
        disp = np.empty((n, self.height, self.width), np.float32) if want_disp else None
        raw = np.empty((n, self.height, self.width), np.int32) if want_raw else None
        self._check(self._lib.sn_infer_batch(self._h, n, x.ctypes.data, _np_ptr(raw), _np_ptr(disp), SN_MEM_HOST, None),
                    "sn_infer_batch")
        if single:
            return (disp[0] if disp is not None else None), (raw[0] if raw is not None else None)
        return disp, raw

    # -- Run (device pointers, e.g. torch tensors' data_ptr(); stream = hipStream_t as int) ------
    def infer_device(self, n: int, in_ptr: int, raw_ptr: int, disp_ptr: int, stream: int = 0):
        self._check(self._lib.sn_infer_batch(self._h, n, in_ptr, raw_ptr or None, disp_ptr or None, SN_MEM_DEVICE,
                                             stream or None), "sn_infer_batch")

    def preprocess_nv12(self, left: np.ndarray, right: np.ndarray, w: int, h: int) -> np.ndarray:
        left = np.ascontiguousarray(left, dtype=np.uint8)
        right = np.ascontiguousarray(right, dtype=np.uint8)
        out = np.empty((6, h, w), np.int8)
        self._check(self._lib.sn_preprocess_nv12(self._h, left.ctypes.data, right.ctypes.data, w, h, out.ctypes.data,
                                                 SN_MEM_HOST, None), "sn_preprocess_nv12")
        return out

    def preprocess_sbs_nv12_device(self, n: int, sbs_ptr: int, out_ptr: int, stream: int = 0):
        """n side-by-side NV12 frames (device, n*3*H*W bytes) -> n int8 model tensors (device), on `stream`."""
        self._check(self._lib.sn_preprocess_sbs_nv12_batch(self._h, n, sbs_ptr, 2 * self.width, self.height, out_ptr,
                                                           SN_MEM_DEVICE, stream or None), "sn_preprocess_sbs_nv12_batch")

    def preprocess_sbs_nv12(self, sbs: np.ndarray) -> np.ndarray:
        """uint8 (n, 3*H*W) side-by-side NV12 frames -> int8 (n, 6, H, W) model tensors (host buffers)."""
        x = np.ascontiguousarray(sbs, dtype=np.uint8).reshape(-1, 3 * self.width * self.height)
        out = np.empty((x.shape[0], 6, self.height, self.width), np.int8)
        self._check(self._lib.sn_preprocess_sbs_nv12_batch(self._h, x.shape[0], x.ctypes.data, 2 * self.width, self.height,
                                                           out.ctypes.data, SN_MEM_HOST, None), "sn_preprocess_sbs_nv12_batch")
        return out

    def infer_sbs_nv12(self, sbs: np.ndarray, want_tensor: bool = False):
        """sbs: uint8 side-by-side NV12 frame (H*3/2 rows of 2W bytes) -> (disp, raw[, tensor])"""
        sbs = np.ascontiguousarray(sbs, dtype=np.uint8)
        disp = np.empty((self.height, self.width), np.float32)
        raw = np.empty((self.height, self.width), np.int32)
        ten = np.empty((6, self.height, self.width), np.int8) if want_tensor else None
        self._check(self._lib.sn_infer_sbs_nv12(self._h, sbs.ctypes.data, 2 * self.width, self.height, raw.ctypes.data,
                                                disp.ctypes.data, _np_ptr(ten), SN_MEM_HOST, None), "sn_infer_sbs_nv12")
        return (disp, raw, ten) if want_tensor else (disp, raw)

    # -- async Run -------------------------------------------------------------------------------
    def submit(self, in6: np.ndarray, raw_out: Optional[np.ndarray], disp_out: Optional[np.ndarray],
               timeout_ms: int = -1) -> int:
        x = np.ascontiguousarray(in6, dtype=np.int8)
        t = C.c_uint64()
        self._check(self._lib.sn_submit(self._h, x.ctypes.data, _np_ptr(raw_out), _np_ptr(disp_out), timeout_ms,
                                        C.byref(t)), "sn_submit")
        return t.value

    def submit_nv12(self, sbs: np.ndarray, raw_out: Optional[np.ndarray], disp_out: Optional[np.ndarray],
                    timeout_ms: int = -1) -> int:
        """async Run on FeedImg's raw side-by-side NV12 frame (uint8, 3*H*W bytes): half the H2D bytes of submit()."""
        x = np.ascontiguousarray(sbs, dtype=np.uint8)
        if x.size != 3 * self.width * self.height:
            raise StereoNetError(-1, "submit_nv12", f"frame has {x.size} bytes, expected {3 * self.width * self.height}")
        t = C.c_uint64()
        self._check(self._lib.sn_submit_nv12(self._h, x.ctypes.data, 2 * self.width, self.height, _np_ptr(raw_out),
                                             _np_ptr(disp_out), timeout_ms, C.byref(t)), "sn_submit_nv12")
        return t.value

    def wait(self, ticket: int) -> float:
        ms = C.c_float()
        self._check(self._lib.sn_wait(self._h, ticket, C.byref(ms)), "sn_wait")
        return ms.value

    def depth_from_raw(self, raw: np.ndarray, focal_px: float = 527.1931762695312, baseline_mm: float = 119.89382172,
                       want_disp: bool = False):
        """Parse()'s dequantisation + depth (parser.cpp:84-86) on the GPU: int32 (H,W) or (n,H,W) -> depth in metres
        (float32, inf where raw == 0) [, disparity px]; bit-identical to the host Parse."""
        r = np.ascontiguousarray(raw, dtype=np.int32)
        # sn_depth_from_raw moves n * H * W elements of the MODEL's size: anything else would run past these arrays
        if r.ndim not in (2, 3) or r.shape[-2:] != (self.height, self.width):
            raise StereoNetError(-1, "depth_from_raw", f"raw shape {r.shape} != ([n,] {self.height}, {self.width})")
        n = 1 if r.ndim == 2 else r.shape[0]
        if n < 1 or n > self.max_batch:
            raise StereoNetError(-1, "depth_from_raw", f"{n} maps, the engine was created for 1..{self.max_batch}")
        depth = np.empty(r.shape, np.float32)
        disp = np.empty(r.shape, np.float32) if want_disp else None
        self._check(self._lib.sn_depth_from_raw(self._h, n, r.ctypes.data, focal_px, baseline_mm, depth.ctypes.data,
                                                _np_ptr(disp), SN_MEM_HOST, None), "sn_depth_from_raw")
        return (depth, disp) if want_disp else depth

    def synchronize(self):
        self._check(self._lib.sn_synchronize(self._h), "sn_synchronize")

    # -- measurement -------------------------------------------------------------------------------
    def set_profiling(self, on: bool):
        self._check(self._lib.sn_set_profiling(self._h, int(on)), "sn_set_profiling")

    def stage_ms(self) -> dict:
        arr = (C.c_float * len(STAGES))()
        self._check(self._lib.sn_get_stage_ms(self._h, arr, len(STAGES)), "sn_get_stage_ms")
        return dict(zip(STAGES, [float(v) for v in arr]))

    def dominant_kernel(self) -> dict:
        name = C.create_string_buffer(128)
        launches = C.c_int()
        fl, by = C.c_double(), C.c_double()
        self._check(self._lib.sn_get_dominant_kernel(self._h, name, 128, C.byref(launches), C.byref(fl), C.byref(by)),
                    "sn_get_dominant_kernel")
        return {"name": name.value.decode(), "launches": launches.value, "flops_per_launch": fl.value,
                "bytes_per_launch": by.value}

    # -- parity hooks --------------------------------------------------------------------------------
    def dbg_conv2d(self, x, wt, bias, k, stride=1, dil=1, lrelu=False, residual=None, x3=False, slots=False, tower32=False, dma=False):
        """dma (with x3 and slots): the kernel of the zero-bordered tensors — k_down_x3s_dma (5x5 stride 2) or
        k_feat_x3s_dma (3x3, also with a residual); the hook also checks that the borders stay zero"""
        x = np.ascontiguousarray(x, np.float32)
        wt = np.ascontiguousarray(wt, np.float32)
        bias = np.ascontiguousarray(bias, np.float32)
        cin, h, w = x.shape
        ho, wo = (h, w) if stride == 1 else (h // 2, w // 2)
        out = np.empty((32, ho, wo), np.float32)
        res = np.ascontiguousarray(residual, np.float32) if residual is not None else None
        self._check(self._lib.sn_dbg_conv2d(self._h, x.ctypes.data, cin, h, w, wt.ctypes.data, bias.ctypes.data, k,
                                            stride, dil, int(lrelu) | (2 if x3 else 0) | (4 if slots else 0) | (8 if tower32 else 0) | (16 if dma else 0), _np_ptr(res),
                                            out.ctypes.data),
                    "sn_dbg_conv2d")
        return out

    def dbg_down0(self, in6, wt, bias, tc=32):
        """in6 int8 (6,h,w) -> float32 (2, 32, ho, wo): first down-conv of both eyes on the fp16 MFMA."""
        x = np.ascontiguousarray(in6, np.int8)
        wt = np.ascontiguousarray(wt, np.float32)
        bias = np.ascontiguousarray(bias, np.float32)
        _, h, w = x.shape
        ho, wo = (h + 15) // 16 * 8, (w + 15) // 16 * 8
        out = np.empty((2, 32, ho, wo), np.float32)
        self._check(self._lib.sn_dbg_down0(self._h, x.ctypes.data, h, w, wt.ctypes.data, bias.ctypes.data, tc,
                                           out.ctypes.data), "sn_dbg_down0")
        return out

    def dbg_down01(self, in6, w0, b0, w1, b1):
        """in6 int8 (6,h,w) -> float32 (2, 32, ho, wo), ho/wo = ceil16/4: the first two down-convs of both eyes as the
        folded 13x13 stride-4 convolution (k_down01_f16 + k_down01_border)."""
        x = np.ascontiguousarray(in6, np.int8)
        w0, b0, w1, b1 = (np.ascontiguousarray(a, np.float32) for a in (w0, b0, w1, b1))
        assert w0.shape == (32, 3, 5, 5) and w1.shape == (32, 32, 5, 5) and b0.shape == (32,) and b1.shape == (32,)
        _, h, w = x.shape
        ho, wo = (h + 15) // 16 * 4, (w + 15) // 16 * 4
        out = np.empty((2, 32, ho, wo), np.float32)
        self._check(self._lib.sn_dbg_down01(self._h, x.ctypes.data, h, w, w0.ctypes.data, b0.ctypes.data, w1.ctypes.data,
                                            b1.ctypes.data, out.ctypes.data), "sn_dbg_down01")
        return out

    def dbg_refin(self, disp_low, in6, dmax, wt, bias, split=False):
        """disp_low float32 (hp/16, wp/16), in6 int8 (6,h,w) -> float32 (32, hp, wp): refinement input conv."""
        x = np.ascontiguousarray(in6, np.int8)
        dl = np.ascontiguousarray(disp_low, np.float32)
        wt = np.ascontiguousarray(wt, np.float32)
        bias = np.ascontiguousarray(bias, np.float32)
        _, h, w = x.shape
        hp, wp = (h + 15) // 16 * 16, (w + 15) // 16 * 16
        assert dl.shape == (hp // 16, wp // 16)
        out = np.empty((32, hp, wp), np.float32)
        self._check(self._lib.sn_dbg_refin(self._h, dl.ctypes.data, x.ctypes.data, h, w, dmax, wt.ctypes.data,
                                           bias.ctypes.data, int(split), out.ctypes.data), "sn_dbg_refin")
        return out

    def dbg_conv3d(self, x, wt, bias, lrelu=False, x3=False, slots=False, dma=False):
        """dma: the aggregation kernel on zero-bordered volumes (k_agg_x3s_dma; needs x3 and slots); the hook also checks that
        the kernel left the borders untouched"""
        x = np.ascontiguousarray(x, np.float32)
        wt = np.ascontiguousarray(wt, np.float32)
        bias = np.ascontiguousarray(bias, np.float32)
        _, d, h, w = x.shape
        out = np.empty_like(x)
        self._check(self._lib.sn_dbg_conv3d(self._h, x.ctypes.data, d, h, w, wt.ctypes.data, bias.ctypes.data,
                                            int(lrelu) | (2 if x3 else 0) | (4 if slots else 0) | (8 if dma else 0), out.ctypes.data),
                    "sn_dbg_conv3d")
        return out

    def dbg_ref_conv_f16(self, x, wt, bias, dil=1, lrelu=False, residual=None, tile_w=0):
        """tile_w: 0 = the width the engine would choose for this launch, 64 / 32 = force that variant (dilation 1, 2)"""
        x = np.ascontiguousarray(x, np.float32)
        wt = np.ascontiguousarray(wt, np.float32)
        bias = np.ascontiguousarray(bias, np.float32)
        _, h, w = x.shape
        out = np.empty((32, h, w), np.float32)
        res = np.ascontiguousarray(residual, np.float32) if residual is not None else None
        flags = int(bool(lrelu)) | (2 if tile_w == 64 else 4 if tile_w == 32 else 0)
        self._check(self._lib.sn_dbg_ref_conv_f16(self._h, x.ctypes.data, h, w, wt.ctypes.data, bias.ctypes.data, dil,
                                                  flags, _np_ptr(res), out.ctypes.data), "sn_dbg_ref_conv_f16")
        return out

    def dbg_ref_conv_f16x3(self, x, wt, bias, dil=1, lrelu=False, residual=None):
        x = np.ascontiguousarray(x, np.float32)
        wt = np.ascontiguousarray(wt, np.float32)
        bias = np.ascontiguousarray(bias, np.float32)
        _, h, w = x.shape
        out = np.empty((32, h, w), np.float32)
        res = np.ascontiguousarray(residual, np.float32) if residual is not None else None
        self._check(self._lib.sn_dbg_ref_conv_f16x3(self._h, x.ctypes.data, h, w, wt.ctypes.data, bias.ctypes.data, dil,
                                                    int(lrelu), _np_ptr(res), out.ctypes.data), "sn_dbg_ref_conv_f16x3")
        return out

    def dbg_ref_block_f16(self, x, w1, b1, w2, b2, dil=1, fused=0):
        """fused: 0 = two conv launches, 2 = row-streaming fused kernel (every dilation: 1, 2, 4, 8)"""
        dil = dil | (fused << 8)
        a = [np.ascontiguousarray(v, np.float32) for v in (x, w1, b1, w2, b2)]
        _, h, w = a[0].shape
        out = np.empty((32, h, w), np.float32)
        self._check(self._lib.sn_dbg_ref_block_f16(self._h, a[0].ctypes.data, h, w, a[1].ctypes.data, a[2].ctypes.data,
                                                   a[3].ctypes.data, a[4].ctypes.data, dil, out.ctypes.data),
                    "sn_dbg_ref_block_f16")
        return out

    def dbg_ref_block_f16x3(self, x, w1, b1, w2, b2, dil=1, streamed=False):
        """the same block on split operands (SN_PREC_F16X3): two k_ref_conv_f16x3 launches, or the row-streaming fused kernel"""
        a = [np.ascontiguousarray(v, np.float32) for v in (x, w1, b1, w2, b2)]
        _, h, w = a[0].shape
        out = np.empty((32, h, w), np.float32)
        self._check(self._lib.sn_dbg_ref_block_f16x3(self._h, a[0].ctypes.data, h, w, a[1].ctypes.data, a[2].ctypes.data,
                                                     a[3].ctypes.data, a[4].ctypes.data, dil, int(streamed), out.ctypes.data),
                    "sn_dbg_ref_block_f16x3")
        return out

    def dbg_ref_tail_f16(self, x, w1, b1, w2, b2, head_w, head_b, low, ups, dnorm, h_out, w_out, form):
        """Last residual block + refinement head on x float32 (n, 32, hk, wk); low (n, hk/ups, wk/ups); form 0 = streamed
        block + k_head_final_f16, 1 = the tail form (one launch).  -> (disp float32, raw int32), each (n, h_out, w_out)."""
        a = [np.ascontiguousarray(v, np.float32) for v in (x, w1, b1, w2, b2, head_w, low)]
        n, _, hk, wk = a[0].shape
        assert a[6].shape == (n, hk // ups, wk // ups)
        disp = np.empty((n, h_out, w_out), np.float32)
        raw = np.empty((n, h_out, w_out), np.int32)
        self._check(self._lib.sn_dbg_ref_tail_f16(self._h, n, a[0].ctypes.data, hk, wk, a[1].ctypes.data, a[2].ctypes.data,
                                                  a[3].ctypes.data, a[4].ctypes.data, a[5].ctypes.data, float(head_b),
                                                  a[6].ctypes.data, ups, float(dnorm), h_out, w_out, form, disp.ctypes.data,
                                                  raw.ctypes.data), "sn_dbg_ref_tail_f16")
        return disp, raw

    def stream_priority_high(self) -> bool:
        """True when the engine's pipeline streams were created with the device's highest priority (SN_STREAM_PRIORITY=1,
        or sn_mgpu_create with more than one device)."""
        return bool(self.dbg_read("stream_prio")[0])

    def dbg_read(self, what: str) -> np.ndarray:
        n = C.c_size_t()
        self._check(self._lib.sn_dbg_read(self._h, what.encode(), None, 0, C.byref(n)), "sn_dbg_read")
        out = np.empty(n.value, np.float32)
        self._check(self._lib.sn_dbg_read(self._h, what.encode(), out.ctypes.data, n.value, C.byref(n)), "sn_dbg_read")
        return out


def mgpu_shard(n: int, ndev: int, k: int):
    """(first, count) of shard k when n pairs are cut over ndev devices (sn_mgpu_shard; needs no GPU)."""
    lib = load_library()
    first, count = C.c_int(), C.c_int()
    rc = lib.sn_mgpu_shard(n, ndev, k, C.byref(first), C.byref(count))
    if rc != 0:
        raise StereoNetError(rc, "sn_mgpu_shard")
    return first.value, count.value


class SnMgpuRing(C.Structure):
    _fields_ = [("next", C.c_uint64), ("slot_ticket", C.c_uint64 * 2)]


class MgpuRing:
    """The ticket / buffer-slot bookkeeping of sn_mgpu_submit_device / sn_mgpu_wait (pure; needs no GPU)."""

    def __init__(self):
        self._lib = load_library()
        self._r = SnMgpuRing()
        self._lib.sn_mgpu_ring_init(C.byref(self._r))

    def submit(self):
        t, s = C.c_uint64(), C.c_int()
        rc = self._lib.sn_mgpu_ring_submit(C.byref(self._r), C.byref(t), C.byref(s))
        if rc != 0:
            raise StereoNetError(rc, "sn_mgpu_ring_submit")
        return t.value, s.value

    def wait(self, ticket: int) -> int:
        s = C.c_int()
        rc = self._lib.sn_mgpu_ring_wait(C.byref(self._r), C.c_uint64(ticket), C.byref(s))
        if rc != 0:
            raise StereoNetError(rc, "sn_mgpu_ring_wait")
        return s.value


class StereoNetMultiGPU:
    """sn_mgpu_*: one batch sharded over the GPUs of one node inside ONE process (one host thread per GPU)."""

    def __init__(self, model_file: str, devices=None, ndev: int = 0, max_batch: int = 1, precision: int = PREC_DEFAULT,
                 refine_chunk: int = 0, piece: int = 0, width: int = 0, height: int = 0, dmax: int = 0):
        self._lib = load_library()
        self._m = C.c_void_p()
        devs = list(devices) if devices is not None else None
        if devs is None and ndev <= 0:      # neither given: every visible GPU (sn_mgpu_create itself rejects ndev <= 0)
            import torch
            ndev = torch.cuda.device_count() if torch.cuda.is_available() else 0
            if ndev <= 0:
                raise StereoNetError(-4, "StereoNetMultiGPU: no GPU visible and neither devices nor ndev given")
        n = len(devs) if devs is not None else ndev
        arr = (C.c_int * n)(*devs) if devs is not None else None
        cfg = SnConfig(-1, max_batch, width, height, dmax, precision, 4, refine_chunk, piece)
        rc = self._lib.sn_mgpu_create(model_file.encode(), C.byref(cfg), arr, n, C.byref(self._m))
        if rc != 0:
            self._m = C.c_void_p()
            raise StereoNetError(rc, f"sn_mgpu_create({model_file!r}, ndev={n})")
        nd, per, kind = C.c_int(), C.c_int(), C.c_int()
        self._lib.sn_mgpu_get_info(self._m, C.byref(nd), C.byref(per), C.byref(kind))
        self.ndev, self.per_device_batch, self.gather_kind = nd.value, per.value, kind.value
        h = C.c_void_p()
        self._lib.sn_mgpu_get_handle(self._m, 0, C.byref(h))
        info = SnIoInfo()
        self._lib.sn_get_io_info(h, C.byref(info))
        self.width, self.height, self.dmax = info.width, info.height, info.dmax
        self.max_batch = max_batch

    def _check(self, rc: int, where: str):
        if rc != 0:
            raise StereoNetError(rc, where, self._lib.sn_mgpu_last_error(self._m).decode() if self._m else "")

    def engine_stream_priority_high(self, k: int = 0) -> bool:
        """True when shard k's engine runs its pipeline on high-priority streams (sn_mgpu_create chooses that itself whenever
        more than one device takes part: its gather runs beside the engines)."""
        h = C.c_void_p()
        self._check(self._lib.sn_mgpu_get_handle(self._m, k, C.byref(h)), "sn_mgpu_get_handle")
        v, n = np.empty(1, np.float32), C.c_size_t()
        rc = self._lib.sn_dbg_read(h, b"stream_prio", v.ctypes.data, 1, C.byref(n))
        if rc != 0:
            raise StereoNetError(rc, "sn_dbg_read(stream_prio)")
        return bool(v[0])

    def infer(self, in6: np.ndarray):
        """int8 (n,6,H,W) host array -> (disp float32 (n,H,W), raw int32 (n,H,W)); the host is the gather root."""
        x = np.ascontiguousarray(in6, dtype=np.int8)
        n = x.shape[0]
        disp = np.empty((n, self.height, self.width), np.float32)
        raw = np.empty((n, self.height, self.width), np.int32)
        self._check(self._lib.sn_mgpu_infer_batch(self._m, n, x.ctypes.data, raw.ctypes.data, disp.ctypes.data),
                    "sn_mgpu_infer_batch")
        return disp, raw

    def infer_device(self, n: int, in_ptrs, raw_root_ptr: int, disp_root_ptr: int):
        """in_ptrs[k] = device pointer of shard k on device k; outputs gathered on device 0."""
        arr = (C.c_void_p * self.ndev)(*[C.c_void_p(p) for p in in_ptrs])
        self._check(self._lib.sn_mgpu_infer_batch_device(self._m, n, arr, raw_root_ptr or None, disp_root_ptr or None),
                    "sn_mgpu_infer_batch_device")

    def submit_device(self, n: int, in_ptrs, raw_root_ptr: int, disp_root_ptr: int) -> int:
        """Asynchronous infer_device: returns a ticket once every device has its shard enqueued (two may be in flight)."""
        arr = (C.c_void_p * self.ndev)(*[C.c_void_p(p) for p in in_ptrs])
        t = C.c_uint64()
        self._check(self._lib.sn_mgpu_submit_device(self._m, n, arr, raw_root_ptr or None, disp_root_ptr or None, C.byref(t)),
                    "sn_mgpu_submit_device")
        return t.value

    def wait(self, ticket: int):
        self._check(self._lib.sn_mgpu_wait(self._m, C.c_uint64(ticket)), "sn_mgpu_wait")

    def close(self):
        if getattr(self, "_m", None) and self._m.value:
            self._lib.sn_mgpu_destroy(self._m)
            self._m = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
