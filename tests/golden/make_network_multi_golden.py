#!/usr/bin/env python3
"""Generates tests/golden/network_multi_golden.npz from the torch restatement (tests/golden/torch_ref.py) for the
HIERARCHICAL refinement (`multi`): seeded inputs + the seeded multi weight blob -> expected
full-resolution disparity and the maps of the coarser levels, plus the two ops only this mode uses (x2 bilinear
upsample, 2x2 average pooling).  Run from the repo root:
    python tests/golden/make_network_multi_golden.py
"""
import hashlib
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(__file__))

import torch_ref  # noqa: E402
from hobot_stereonet_amd import spec, synth, weights  # noqa: E402

torch.set_num_threads(8)
CASES = [("c96x64_d48", 96, 64, 48, 3), ("c160x96_d96", 160, 96, 96, 4), ("c100x52_d32", 100, 52, 32, 5)]


def sha(a: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def main():
    blob = weights.synthetic(0, spec.MULTI_LEVELS)
    out = {"weights_sha256": np.array(sha(blob))}
    for name, w, h, d, seed in CASES:
        x = synth.model_input_i8(w, h, d, seed)
        r = torch_ref.forward(blob, x, d)
        out[f"{name}.input_sha256"] = np.array(sha(x))
        out[f"{name}.disp"] = r["disp"].astype(np.float32)
        out[f"{name}.disp_low"] = r["disp_low"].astype(np.float32)
        # r["levels"] = maps of levels 3, 2, 1, 0 in that order; keep the coarse ones under their level number
        for i, m in enumerate(r["levels"][:-1]):
            out[f"{name}.level{spec.MULTI_LEVELS - 1 - i}"] = m.astype(np.float32)
        print(name, "disp mean", r["disp"].mean(), "levels", [m.shape for m in r["levels"]])
    rng = np.random.default_rng(11)
    lo = rng.standard_normal((5, 7)).astype(np.float32)
    out["op.up2.x"] = lo
    out["op.up2.y"] = (F.interpolate(torch.from_numpy(lo)[None, None], scale_factor=2, mode="bilinear", align_corners=False) * 2.0)[0, 0].numpy()
    im = rng.standard_normal((3, 8, 12)).astype(np.float32)
    out["op.pool.x"] = im
    out["op.pool.y"] = F.avg_pool2d(torch.from_numpy(im)[None], 2)[0].numpy()
    path = os.path.join(os.path.dirname(__file__), "network_multi_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
