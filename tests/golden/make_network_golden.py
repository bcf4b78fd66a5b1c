#!/usr/bin/env python3
"""Generates tests/golden/network_golden.npz from the torch restatement
(tests/golden/torch_ref.py): seeded inputs -> expected outputs for the whole
SN-K4 path at two small sizes, plus per-op vectors.  Run from the repo root:
    python tests/golden/make_network_golden.py
Inputs/weights are regenerated from seeds at test time (synth.py / weights.py);
their checksums are stored so that drift in the generators is detected.
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
from hobot_stereonet_amd import synth, weights  # noqa: E402

torch.set_num_threads(8)
CASES = [("c96x64_d48", 96, 64, 48, 3), ("c160x96_d96", 160, 96, 96, 4), ("c100x52_d32", 100, 52, 32, 5)]


def sha(a: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def main():
    blob = weights.synthetic(0)
    out = {"weights_sha256": np.array(sha(blob))}
    for name, w, h, d, seed in CASES:
        x = synth.model_input_i8(w, h, d, seed)
        r = torch_ref.forward(blob, x, d)
        out[f"{name}.input_sha256"] = np.array(sha(x))
        out[f"{name}.disp"] = r["disp"].astype(np.float32)
        out[f"{name}.disp_low"] = r["disp_low"].astype(np.float32)
        out[f"{name}.cost"] = r["cost"].astype(np.float32)
        print(name, "disp mean", r["disp"].mean(), "low range", r["disp_low"].min(), r["disp_low"].max())
    # per-op vectors
    rng = np.random.default_rng(7)
    x = rng.standard_normal((5, 13, 17)).astype(np.float32)
    for tag, k, s, p, dil, co in [("k3", 3, 1, 1, 1, 4), ("k3d4", 3, 1, 4, 4, 3), ("k5s2", 5, 2, 2, 1, 6)]:
        wt = rng.standard_normal((co, 5, k, k)).astype(np.float32)
        b = rng.standard_normal(co).astype(np.float32)
        y = F.conv2d(torch.from_numpy(x)[None], torch.from_numpy(wt), torch.from_numpy(b), stride=s, padding=p, dilation=dil)
        out[f"op.conv2d.{tag}.w"] = wt
        out[f"op.conv2d.{tag}.b"] = b
        out[f"op.conv2d.{tag}.y"] = y[0].numpy()
    out["op.conv2d.x"] = x
    x3 = rng.standard_normal((3, 4, 6, 7)).astype(np.float32)
    w3 = rng.standard_normal((2, 3, 3, 3, 3)).astype(np.float32)
    b3 = rng.standard_normal(2).astype(np.float32)
    out["op.conv3d.x"], out["op.conv3d.w"], out["op.conv3d.b"] = x3, w3, b3
    out["op.conv3d.y"] = F.conv3d(torch.from_numpy(x3)[None], torch.from_numpy(w3), torch.from_numpy(b3), padding=1)[0].numpy()
    lo = rng.standard_normal((5, 7)).astype(np.float32)
    out["op.up.x"] = lo
    out["op.up.y"] = (F.interpolate(torch.from_numpy(lo)[None, None], scale_factor=16, mode="bilinear", align_corners=False) * 16.0)[0, 0].numpy()
    cost = (rng.standard_normal((6, 5, 9)) * 3).astype(np.float32)
    out["op.sam.x"] = cost
    out["op.sam.y"] = torch_ref.soft_argmin(torch.from_numpy(cost)[None])[0].numpy()
    path = os.path.join(os.path.dirname(__file__), "network_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
