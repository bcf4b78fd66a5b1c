#!/usr/bin/env python3
"""A/B of an environment switch read at sn_create (default SN_TAIL_FUSE): the two forms must produce the same maps.
    python scripts/tail_ab.py [ENVVAR]   -> prints max |d raw|, max |d disp| per geometry and the EPE of both forms vs the oracle"""
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
from hobot_stereonet_amd import api, spec, synth, weights  # noqa: E402

VAR = sys.argv[1] if len(sys.argv) > 1 else "SN_TAIL_FUSE"
CASES = [(200, 120, 64, 5, 1), (160, 96, 96, 3, 4), (1280, 720, 192, 5, 1), (1242, 375, 256, 3, 4), (1242, 375, 256, 9, 1), (64, 48, 32, 2, 1),
         (330, 250, 48, 3, 1)]
import oracle_py  # noqa: E402

ok = True
for (w, h, d, n, levels) in CASES:
    blob = weights.synthetic(0, levels)
    xs = np.stack([synth.model_input_i8(w, h, d, 70 + i) for i in range(n)])
    outs = {}
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "m.snw")
        weights.save_snw(path, blob, w, h, d)
        for v in ("0", "1"):
            os.environ[VAR] = v
            with api.StereoNetHIP(path, device=0, max_batch=n) as eng:
                outs[v] = eng.infer(xs)
    dr = int(np.abs(outs["0"][1].astype(np.int64) - outs["1"][1]).max())
    dd = float(np.abs(outs["0"][0] - outs["1"][0]).max())
    line = f"{w}x{h} D={d} n={n} levels={levels}: max|d raw|={dr} max|d disp|={dd:.3e}"
    if w * h <= 400 * 300:
        od = oracle_py.forward(blob, xs[0], d)[0]
        line += f"  EPE0={np.abs(outs['0'][0][0] - od).mean():.3e} EPE1={np.abs(outs['1'][0][0] - od).mean():.3e}"
    print(line, flush=True)
    ok = ok and dd < 1e-3
sys.exit(0 if ok else 1)
