#!/usr/bin/env python3
"""Where does EPE <= 1e-3 px hold?    Every parity test and the bench use ONE weight file —
weights.synthetic(0), with the refinement head's gain hand-set so that D * r is of the order of a pixel.  The model the
reference runs is unknown (stereonet_infer/src/stereonet_node.cpp:131-136 only checks that `model_file` exists), so the
bound north_star states has to be shown as a property of the KERNELS over a plausible envelope of weights:

    weight seeds 0..7  x  refinement-head gain {1, 2, 4, 8}  x  low-resolution activation scale {0.5, 1, 2}
    at C1 (960x540 D=48), C2 (1280x720 D=192, single-scale) and C5 (1242x375 D=256, hierarchical), modes F16 / F16rne / F16X3 / FP32

Per cell: mean and max |disp - oracle| in px (HIP path through the C ABI vs oracle/stereonet_oracle.c on the same input and
weights), plus what the envelope means in pixels: `refine_px` = mean |disp - upsampled soft-argmin map| of the oracle.

    python scripts/epe_sensitivity.py [--quick] [--seeds N] > profiles/r05_epe_sensitivity.txt
"""
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle_py  # noqa: E402
from hobot_stereonet_amd import api, synth, weights  # noqa: E402

quick = "--quick" in sys.argv
nseeds = int(sys.argv[sys.argv.index("--seeds") + 1]) if "--seeds" in sys.argv else (2 if quick else 8)
which = sys.argv[sys.argv.index("--config") + 1] if "--config" in sys.argv else "all"
GAINS = (1.0, 8.0) if quick else (1.0, 2.0, 4.0, 8.0)
ACTS = (1.0,) if quick else (0.5, 1.0, 2.0)
# F16 = the library default (sum-preserving rounding of the tower's 3x3 weights); F16rne = the same mode with plain
# round-to-nearest weights (SN_W_ROUND=rne): what rounds 1-4 shipped
MODES = (("F16", api.PREC_F16, None), ("F16rne", api.PREC_F16, "rne"), ("F16X3", api.PREC_F16X3, None), ("FP32", api.PREC_FP32, None))
CONFIGS = tuple(c for c in (("C1 960x540 D=48 single", 960, 540, 48, 1), ("C2 1280x720 D=192 single", 1280, 720, 192, 1),
                            ("C5 1242x375 D=256 multi", 1242, 375, 256, 4))
                if which == "all" or c[0].lower().startswith(which.lower()))
BOUND = {"F16": 1e-3, "F16rne": 1e-3, "F16X3": 2e-4, "FP32": 2e-4}

oracle_py.build()
print("# EPE = mean |disp - oracle| in px; bound 1e-3 (north_star) for F16, 2e-4 asked of F16X3 / FP32")
print("# refine_px = mean |oracle disp - x16 upsample of its soft-argmin map|: what the refinement adds, in pixels")
print("# cells: every head gain at activation scale 1, every activation scale at head gain 1 (the fp16 tower's error is")
print("# linear in the head gain and indifferent to the activation scale: the rows show both)")
print(f"# seeds 0..{nseeds - 1}, head gain {GAINS}, activation scale {ACTS}; oracle on {oracle_py.num_threads()} host threads")
td = tempfile.mkdtemp(prefix="sn_sens_")
t_start = time.time()
worst = {}          # (config, mode, gain) -> (epe, seed, act)
rows = 0
for cname, w, h, d, levels in CONFIGS:
    print(f"\n## {cname}")
    print(f"{'seed':>4} {'gain':>4} {'act':>4} {'refine_px':>9} " + " ".join(f"{m + ' mean':>11} {m + ' max':>10}" for m, _, _ in MODES))
    for seed in range(nseeds):
        x = synth.model_input_i8(w, h, d, 500 + seed)
        for gain, act in [(g, 1.0) for g in GAINS] + [(1.0, a) for a in ACTS if a != 1.0]:
                blob = weights.synthetic(seed, levels, head_gain=gain, act_scale=act)
                path = os.path.join(td, "m.snw")
                weights.save_snw(path, blob, w, h, d)
                od, _, olow = oracle_py.forward(blob, x, d)
                up = oracle_py.upsample_bilinear(olow, 16, 16.0)[:h, :w]
                refine_px = float(np.abs(od - up).mean())
                cells = []
                for mname, prec, wround in MODES:
                    if wround:
                        os.environ["SN_W_ROUND"] = wround
                    else:
                        os.environ.pop("SN_W_ROUND", None)
                    with api.StereoNetHIP(path, device=0, precision=prec) as eng:
                        disp, _ = eng.infer(x)
                    os.environ.pop("SN_W_ROUND", None)
                    err = np.abs(disp - od)
                    e_mean, e_max = float(err.mean()), float(err.max())
                    cells.append(f"{e_mean:11.3e} {e_max:10.3e}")
                    key = (cname, mname, gain)
                    if key not in worst or e_mean > worst[key][0]:
                        worst[key] = (e_mean, seed, act, refine_px)
                print(f"{seed:4d} {gain:4.0f} {act:4.1f} {refine_px:9.3f} " + " ".join(cells), flush=True)
                rows += 1

print(f"\n## worst mean EPE per (config, mode, head gain) over seeds x activation scales   [{rows} cells, {time.time() - t_start:.0f} s]")
print(f"{'config':<26} {'mode':<6} {'gain':>4} {'worst EPE':>10} {'at seed':>7} {'act':>4} {'refine_px':>9}  within bound")
for (cname, mname, gain), (e, seed, act, rpx) in sorted(worst.items()):
    ok = e < BOUND[mname]
    print(f"{cname:<26} {mname:<6} {gain:4.0f} {e:10.3e} {seed:7d} {act:4.1f} {rpx:9.3f}  {'yes' if ok else 'NO'} ({BOUND[mname]:.0e})")
print("\n# F16 envelope: the largest head gain at which EVERY seed / activation scale stays below 1e-3 px")
for cname, *_ in CONFIGS:
    for mode in ("F16", "F16rne"):
        good = [g for g in GAINS if worst[(cname, mode, g)][0] < 1e-3]
        bad = [g for g in GAINS if g not in good]
        lim = max([g for g in good if all(b > g for b in bad)] or [0])
        rpx = worst[(cname, mode, lim)][3] if lim else 0.0
        print(f"{cname:<26} {mode:<6} holds up to head gain {lim:.0f}" + (f" (refinement ~{rpx:.2f} px mean at its worst seed)" if lim else " (not even at gain 1)"))
    print(f"{cname:<26} F16X3 / FP32 hold everywhere: "
          f"{all(worst[(cname, m, g)][0] < BOUND[m] for m in ('F16X3', 'FP32') for g in GAINS)}")
