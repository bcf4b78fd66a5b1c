#!/usr/bin/env python3
"""Where does the fp16 tower's error come from?  (follow-up of scripts/epe_sensitivity.py.)
CPU emulation of the refinement tower (torch, float64 reference) with the fp16 roundings of SN_PREC_F16 switched on
ONE GROUP AT A TIME:
    W  the 3x3 weights of the twelve tower convs, rounded to nearest once at model load (rounds 1-4)
    w  the same weights with the library's sum-preserving rounding of every 3x3 kernel (round 5, api.round_kernels_f16)
    I  the output of ref.in (stored fp16)
    T  the intermediate of every residual block (t, fp16 in LDS)
    S  the block outputs = the residual stream (stored fp16), blocks 0..4
    L  the last block's output (what the head contracts)
and all of them together (= what the HIP path computes up to summation order).  Output: mean |D * (r - r_ref)| in px per
group, for several weight seeds.  No GPU, no oracle: the tower alone on a random start map and image.

    python scripts/f16_error_sources.py [--seeds 0,3,6] [--size 384x256] > profiles/r05_f16_error_sources.txt
"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hobot_stereonet_amd import api, spec, weights  # noqa: E402

torch.set_num_threads(8)
arg = lambda k, d: sys.argv[sys.argv.index(k) + 1] if k in sys.argv else d
seeds = [int(s) for s in arg("--seeds", "0,3,6").split(",")]
W_, H_ = (int(v) for v in arg("--size", "384x256").split("x"))
D = 192


def q(x, on):
    return x.to(torch.float16).to(x.dtype) if on else x


def qw(w, rnd):
    if "W" in rnd:
        return q(w, True)
    if "w" in rnd:
        return torch.from_numpy(api.round_kernels_f16(w.numpy().astype(np.float32)).astype(np.float64))
    return w


def lrelu(x):
    return F.leaky_relu(x, spec.LRELU_SLOPE)


def tower(blob, x4, rnd):
    g = lambda n: torch.from_numpy(weights.tensor(blob, n).copy()).double()
    x = lrelu(F.conv2d(x4, g("ref.in.w"), g("ref.in.b"), padding=1))
    x = q(x, "I" in rnd)
    nb = len(spec.REF_DILATIONS)
    for i, dil in enumerate(spec.REF_DILATIONS):
        w1, w2 = qw(g(f"ref.res{i}.1.w"), rnd), qw(g(f"ref.res{i}.2.w"), rnd)
        t = q(lrelu(F.conv2d(x, w1, g(f"ref.res{i}.1.b"), padding=dil, dilation=dil)), "T" in rnd)
        y = lrelu(x + F.conv2d(t, w2, g(f"ref.res{i}.2.b"), padding=dil, dilation=dil))
        x = q(y, ("L" if i == nb - 1 else "S") in rnd)
    return F.conv2d(x, g("ref.out.w"), g("ref.out.b"), padding=1)


print(f"# fp16 rounding groups of the refinement tower, one at a time: mean |D (r - r_ref)| in px, D = {D}, {W_}x{H_}")
print("# 'W mean' / 'w mean' = SIGNED mean of that group's error: the weight term is an offset, the sum-preserving rounding removes it")
print(f"{'seed':>4} {'|D r|':>7} " + " ".join(f"{n:>9}" for n in ("W", "W mean", "w", "w mean", "I", "T", "S", "L", "all (W)", "all (w)")))
for seed in seeds:
    blob = weights.synthetic(seed)
    rng = np.random.default_rng(900 + seed)
    img = torch.from_numpy(rng.integers(-128, 128, (1, 3, H_, W_)).astype(np.float64) / 128.0)
    low = torch.from_numpy(rng.random((1, 1, H_ // 16, W_ // 16)) * (D / 16.0))
    up = F.interpolate(low, scale_factor=16, mode="bilinear", align_corners=False) * 16.0
    x4 = torch.cat([up / D, img], 1)
    with torch.no_grad():
        ref = tower(blob, x4, "")
        errs, means = {}, {}
        for grp in ("W", "w", "I", "T", "S", "L", "WITSL", "wITSL"):
            e = D * (tower(blob, x4, grp) - ref)
            errs[grp], means[grp] = float(e.abs().mean()), float(e.mean())
    cols = [f"{errs['W']:9.2e}", f"{means['W']:+9.2e}", f"{errs['w']:9.2e}", f"{means['w']:+9.2e}"] + [f"{errs[g]:9.2e}" for g in ("I", "T", "S", "L", "WITSL", "wITSL")]
    print(f"{seed:4d} {float((D * ref).abs().mean()):7.3f} " + " ".join(cols), flush=True)
