#!/usr/bin/env python3
"""Where SN_PREC_AUTO's envelope comes from, and what AUTO then does.

Per shape class, weight seed 0..7 x refinement-head gain {1, 2, 4, 8}:
    residual_px   the library's own statistic (sn_get_refine_stats): sum_k 2^k mean |D_k r_k| of an SN_PREC_F16 call
    level_px      the per-level means behind it
    F16 / F16X3   mean |disp - oracle| in px of the two forced modes
    F16 vs X3     mean |F16 - F16X3|: what the self-check measures (no oracle in the product)
    AUTO          the mode an SN_PREC_AUTO handle ended in for this call, its EPE vs the oracle, reruns

    python scripts/auto_envelope.py [--config c2|c5|c1|c2m] [--seeds N] > profiles/r06_auto_envelope_<config>.txt
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

nseeds = int(sys.argv[sys.argv.index("--seeds") + 1]) if "--seeds" in sys.argv else 8
which = sys.argv[sys.argv.index("--config") + 1] if "--config" in sys.argv else "all"
GAINS = (1.0, 2.0, 4.0, 8.0)
CONFIGS = {"c1": ("C1 960x540 D=48 single", 960, 540, 48, 1), "c2": ("C2 1280x720 D=192 single", 1280, 720, 192, 1),
           "c5": ("C5 1242x375 D=256 multi", 1242, 375, 256, 4), "c2m": ("C2 1280x720 D=192 multi", 1280, 720, 192, 4),
           "c5s": ("C5 1242x375 D=256 single", 1242, 375, 256, 1)}
todo = [CONFIGS[k] for k in (CONFIGS if which == "all" else which.split(","))]

oracle_py.build()
print("# residual_px = sn_get_refine_stats of an SN_PREC_F16 call: sum_k 2^k mean |D_k r_k| (full-resolution pixels the refinement adds)")
print("# EPE = mean |disp - oracle| px; 'F16 vs X3' = mean |F16 - F16X3| px (the self-check's measurement); slope = F16 EPE / residual_px")
print(f"# seeds 0..{nseeds - 1}, head gain {GAINS}; oracle on {oracle_py.num_threads()} host threads")
td = tempfile.mkdtemp(prefix="sn_env_")
t0 = time.time()
for cname, w, h, d, levels in todo:
    print(f"\n## {cname}   envelope_px = {api.load_library().sn_auto_envelope_px(levels):.3f}")
    print(f"{'seed':>4} {'gain':>4} {'residual_px':>11} {'level_px (0..)':<34} {'F16 EPE':>10} {'F16X3 EPE':>10} {'F16 vs X3':>10} "
          f"{'slope':>9} | {'AUTO':>6} {'AUTO EPE':>10} {'reruns':>6} {'limit_px':>8} {'ok':>3}")
    worst_slope, max_ok_res, min_bad_res, bad_auto = 0.0, 0.0, 1e30, 0
    for seed in range(nseeds):
        x = synth.model_input_i8(w, h, d, 500 + seed)
        for gain in GAINS:
            blob = weights.synthetic(seed, levels, head_gain=gain)
            path = os.path.join(td, "m.snw")
            weights.save_snw(path, blob, w, h, d)
            od = oracle_py.forward(blob, x, d)[0]
            with api.StereoNetHIP(path, device=0, precision=api.PREC_F16) as eng:
                d16, _ = eng.infer(x)
                st = eng.refine_stats()
            with api.StereoNetHIP(path, device=0, precision=api.PREC_F16X3) as eng:
                dx3, _ = eng.infer(x)
                stx = eng.refine_stats()
            with api.StereoNetHIP(path, device=0, precision=api.PREC_AUTO) as eng:
                da, _ = eng.infer(x)
                sta = eng.refine_stats()
            e16, ex3, eab, ea = (float(np.abs(a - b).mean()) for a, b in ((d16, od), (dx3, od), (d16, dx3), (da, od)))
            res = st["residual_px"]
            slope = e16 / res if res > 0 else 0.0
            worst_slope = max(worst_slope, slope)
            if e16 < 1e-3:
                max_ok_res = max(max_ok_res, res)
            else:
                min_bad_res = min(min_bad_res, res)
            ok = ea < 1e-3
            bad_auto += 0 if ok else 1
            # the statistic must not depend on the arithmetic (same maps up to the modes' distance)
            assert abs(stx["residual_px"] - res) <= 0.02 * res + 1e-3, (stx["residual_px"], res)
            lv = " ".join(f"{v:7.3f}" for v in st["level_px"])
            print(f"{seed:4d} {gain:4.0f} {res:11.4f} {lv:<34} {e16:10.3e} {ex3:10.3e} {eab:10.3e} {slope:9.2e} | "
                  f"{sta['precision_last']:>6} {ea:10.3e} {sta['reruns']:6d} {sta['limit_px']:8.3f} {'yes' if ok else 'NO':>3}", flush=True)
    print(f"# {cname}: worst F16 slope {worst_slope:.3e} px per px; largest residual with F16 EPE < 1e-3: {max_ok_res:.3f} px; "
          f"smallest residual with F16 EPE >= 1e-3: {min_bad_res if min_bad_res < 1e29 else float('nan'):.3f} px; "
          f"AUTO cells over 1e-3: {bad_auto}   [{time.time() - t0:.0f} s]")
