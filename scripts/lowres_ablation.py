#!/usr/bin/env python3
"""Per-layer precision ablation of the low-resolution branch: which of its 20 split-operand layers
(down-convs 1-3, the thirteen 3x3 feature convs f0..f12, aggregation layers agg0..agg3) could run on fewer than three fp16
MFMAs per product.  The library's SN_ABLATE_W / SN_ABLATE_X switches reproduce exactly what such a kernel would compute
(an MFMA with a zeroed operand adds exact zeros):
    w = weights rounded to fp16        (drops xh*wl: 2 MFMAs)
    x = input activations rounded      (drops xl*wh: 2 MFMAs)
    p = both                           (plain fp16: 1 MFMA)
Prints EPE vs the CPU oracle (px, mean |disp - oracle|) per (layer, mode) at C2 (1280x720 D=192) and C5-multi
(1242x375 D=256, hierarchical), then whole-branch rows at the ten seeded random geometries of tests/test_gpu_multi.py.
    python scripts/lowres_ablation.py [--quick] > profiles/r04_lowres_ablation.txt"""
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
os.environ["SN_AGG_DMA"] = "0"          # plain split-slot layouts throughout (bit-identical to the zero-bordered ones), so
os.environ["SN_DOWN_DMA"] = "0"         # that every row of the table runs the same kernels
os.environ["SN_DOWN01"] = "0"           # down-convs 0 and 1 as their own layers: folded (the default since round 5) there is no
                                        # down1 input tensor / weight set for the "down1" rows to ablate
os.environ["SN_PRECISION"] = "f16"      # the table is about the fp16 mode's layers (the library default is SN_PREC_AUTO)
import oracle_py  # noqa: E402
from hobot_stereonet_amd import api, build as snbuild, synth, weights  # noqa: E402

# the ablation switches exist in the diagnostic build only (-DSN_DIAGNOSTICS=1), never in the shipping library
if not os.path.exists(snbuild.LIB_DIAG) or os.path.getmtime(snbuild.LIB_DIAG) < os.path.getmtime(snbuild.LIB):
    snbuild.build_diag()
os.environ["STEREONET_HIP_LIB"] = snbuild.LIB_DIAG

LAYERS = ["down1", "down2", "down3"] + [f"f{i}" for i in range(13)] + [f"agg{i}" for i in range(4)]
quick = "--quick" in sys.argv


def run(path, x, n=1, **env):
    for k in ("SN_ABLATE_W", "SN_ABLATE_X"):
        os.environ.pop(k, None)
    os.environ.update(env)
    with api.StereoNetHIP(path, device=0, max_batch=n) as eng:
        return eng.infer(x)[0]


def cases():
    out = [("C2 1280x720 D=192 single", 1280, 720, 192, 1, 300), ("C5 1242x375 D=256 multi", 1242, 375, 256, 4, 400)]
    rng = np.random.default_rng(2026)
    for i in range(10):
        w = int(rng.integers(34, 330))
        h = int(rng.integers(18, 200))
        d = int(rng.choice([16, 32, 48, 64, 96, 128, 192, 256]))
        d = min(d, max(16, (w // 2) // 16 * 16))
        int(rng.integers(1, 6))
        out.append((f"random {w}x{h} D={d} {'multi' if i % 2 else 'single'}", w, h, d, 4 if i % 2 else 1, 700))
    return out


print("# EPE vs the CPU oracle in px; bar 1e-3 (north_star), adoption bar of VERDICT r3 item 3: 6e-4 everywhere")
td = tempfile.mkdtemp()
for ci, (name, w, h, d, levels, seed) in enumerate(cases()):
    blob = weights.synthetic(0, levels)
    path = os.path.join(td, f"m{ci}.snw")
    weights.save_snw(path, blob, w, h, d)
    x = synth.model_input_i8(w, h, d, seed)
    od = oracle_py.forward(blob, x, d)[0]
    epe = lambda disp: float(np.abs(disp - od).mean())
    base = epe(run(path, x))
    allw = epe(run(path, x, SN_ABLATE_W="all"))
    allx = epe(run(path, x, SN_ABLATE_X="all"))
    allp = epe(run(path, x, SN_ABLATE_W="all", SN_ABLATE_X="all"))
    print(f"\n== {name}: 3 MFMAs everywhere (the product) {base:.3e} | whole branch: w {allw:.3e}  x {allx:.3e}  p {allp:.3e}", flush=True)
    if ci >= 2 or quick and ci >= 1:
        continue
    print(f"{'layer':>6s} {'w (2 MFMAs)':>12s} {'x (2 MFMAs)':>12s} {'p (1 MFMA)':>12s}")
    for L in LAYERS:
        ew = epe(run(path, x, SN_ABLATE_W=L))
        ex = epe(run(path, x, SN_ABLATE_X=L))
        ep = epe(run(path, x, SN_ABLATE_W=L, SN_ABLATE_X=L))
        print(f"{L:>6s} {ew:12.3e} {ex:12.3e} {ep:12.3e}", flush=True)
