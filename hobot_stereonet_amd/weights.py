""".snw weight file — plays the role of ``hobot_stereonet.hbm``: it is what the
unchanged ``model_file`` ROS parameter points at (the reference only checks that
the file exists: stereonet_infer/src/stereonet_node.cpp:131-134) and what
``sn_create`` loads.

Layout (little endian):
  0   char[4]  magic "SNW1"
  4   u32      version (1)
  8   u32      model input width   (GetModelInputSize, stereonet_node.cpp:45)
  12  u32      model input height
  16  u32      max disparity D (multiple of 16; 192 = 16*12, parser.cpp:86)
  20  u32      channels C (32)
  24  u32      n_down, 28 u32 n_feat_res, 32 u32 n_agg, 36 u32 n_ref_res
  40  u32[6]   refinement dilations
  64  u64      n_params
  72  u64      refinement levels (0 or 1 = single-scale tower; 4 = hierarchical "multi": the towers of levels 1..3
               follow the single-scale network in the blob, so n_params = spec.param_count(levels))
  80  f32[n_params]  tensors in spec.layers(levels) order, PyTorch layouts
There is no trained StereoNet checkpoint available offline; ``synthetic()`` makes
seeded random weights scaled so activations stay O(1) through both towers.
"""
from __future__ import annotations

import struct

import numpy as np

from . import spec

MAGIC = b"SNW1"
HEADER_BYTES = 80


def synthetic(seed: int = 0, levels: int = 1, head_gain: float = 1.0, act_scale: float = 1.0) -> np.ndarray:
    """Seeded random weights, flat float32 blob in canonical order.  The first spec.param_count() values do not
    depend on `levels` (a multi blob starts with the single blob of the same seed).

    head_gain / act_scale span the envelope in which the precision modes are characterised (scripts/epe_sensitivity.py,
    profiles/r05_epe_sensitivity.txt): head_gain multiplies the refinement heads (`ref*.out`, weights and bias), i.e. the
    residual D * r every refinement level adds — 1 gives ~0.7 px mean |D r| at D = 192, 8 gives ~6 px; act_scale multiplies
    the first feature layer (`feat.down0`), which scales every activation of the low-resolution branch and with it the
    matching costs: the soft-argmin goes from flat (0.5) to near one-hot (2).  Both default to 1: the blob every other
    test and the bench use."""
    rng = np.random.default_rng(seed)
    parts = []
    he = lambda fan_in: np.sqrt(2.0 / (1.0 + spec.LRELU_SLOPE ** 2) / fan_in)
    for l in spec.layers(levels):
        std = he(l.fan_in)
        bias_std = 0.05
        if ".res" in l.name and l.name.endswith(".2"):
            std *= 0.5                       # keep the residual towers from growing
        elif l.name == "agg.out":
            std = 2.0 / np.sqrt(l.fan_in)    # cost spread: soft-argmin neither flat nor one-hot
        elif l.name.startswith("ref") and l.name.endswith(".out"):
            std = 0.004 / np.sqrt(l.fan_in)  # D * r of the order of a pixel
            bias_std = 0.0005
        w = rng.standard_normal(l.w_numel).astype(np.float32) * np.float32(std)
        b = (rng.standard_normal(l.b_numel) * bias_std).astype(np.float32)
        gain = head_gain if (l.name.startswith("ref") and l.name.endswith(".out")) else act_scale if l.name == "feat.down0" else 1.0
        if gain != 1.0:
            w = w * np.float32(gain)
            b = b * np.float32(gain)
        parts.append(w)
        parts.append(b)
    blob = np.concatenate(parts).astype(np.float32)
    assert blob.size == spec.param_count(levels)
    return blob


def save_snw(path: str, blob: np.ndarray, w: int = spec.DEFAULT_W, h: int = spec.DEFAULT_H,
             d: int = spec.DEFAULT_D) -> None:
    blob = np.ascontiguousarray(blob, dtype=np.float32)
    levels = spec.levels_of(blob.size)
    if d % 16 or d < 16:
        raise ValueError("D must be a positive multiple of 16")
    hdr = MAGIC + struct.pack("<9I", 1, w, h, d, spec.C, spec.N_DOWN, spec.N_FEAT_RES,
                              spec.N_AGG, spec.N_REF_RES)
    hdr += struct.pack("<6I", *spec.REF_DILATIONS)
    hdr += struct.pack("<QQ", blob.size, 0 if levels == 1 else levels)
    assert len(hdr) == HEADER_BYTES
    with open(path, "wb") as f:
        f.write(hdr)
        f.write(blob.tobytes())


def load_snw(path: str):
    """-> (blob float32, dict(width, height, dmax, levels))"""
    with open(path, "rb") as f:
        hdr = f.read(HEADER_BYTES)
        if len(hdr) != HEADER_BYTES or hdr[:4] != MAGIC:
            raise ValueError(f"{path}: not an SNW1 file")
        ver, w, h, d, c, nd, nfr, na, nrr = struct.unpack("<9I", hdr[4:40])
        dil = struct.unpack("<6I", hdr[40:64])
        n, lv = struct.unpack("<QQ", hdr[64:80])
        lv = 1 if lv == 0 else int(lv)
        if (ver, c, nd, nfr, na, nrr) != (1, spec.C, spec.N_DOWN, spec.N_FEAT_RES, spec.N_AGG,
                                          spec.N_REF_RES) or tuple(dil) != spec.REF_DILATIONS:
            raise ValueError(f"{path}: architecture header does not match SN-K4")
        if lv not in (1, spec.MULTI_LEVELS) or n != spec.param_count(lv):
            raise ValueError(f"{path}: {n} parameters do not match {lv} refinement level(s)")
        blob = np.frombuffer(f.read(4 * n), dtype=np.float32)
        if blob.size != n:
            raise ValueError(f"{path}: truncated weight blob")
    return blob.copy(), {"width": w, "height": h, "dmax": d, "levels": lv}


def tensor(blob: np.ndarray, name: str) -> np.ndarray:
    off, shape = spec.offsets(spec.levels_of(blob.size))[name]
    n = int(np.prod(shape))
    return blob[off:off + n].reshape(shape)
