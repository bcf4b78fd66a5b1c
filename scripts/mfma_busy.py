#!/usr/bin/env python3
"""Matrix-pipe utilisation per kernel from ONE rocprofv3 pass with `--pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE`
(plus --kernel-trace only: SQ and GRBM counters use different slot pools, so they fit one pass).
SQ_VALU_MFMA_BUSY_CYCLES is the sum over all SIMDs of the cycles their matrix pipe was busy (checked: exactly
32 x the number of v_mfma_f32_32x32x16_f16 a tower launch issues).  Reported per kernel:
    mfma_busy = BUSY / (SIMDS x duration x 2.4 GHz),   SIMDS = 256 CUs x 4
i.e. the fraction of the matrix pipes' peak-clock time (= achieved dense MFMA FLOP/s / peak FLOP/s for fp16 32x32x16).
GRBM_GUI_ACTIVE is kept raw: this rocprofv3 sums it over ~10 counter instances (GUI_ACTIVE / duration = 21 "GHz"), so
it is only used as a ratio between kernels, not as a cycle count.

    python scripts/mfma_busy.py <counter_collection.csv> <out.json>
"""
import collections
import csv
import json
import sys

SIMDS = 256 * 4
agg = collections.defaultdict(lambda: {"n": 0, "busy": 0.0, "active": 0.0, "ns": 0.0})
seen = set()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")
    a = agg[k]
    v = float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_VALU_MFMA_BUSY_CYCLES":
        a["busy"] += v
    elif r["Counter_Name"] == "GRBM_GUI_ACTIVE":
        a["active"] += v
    did = r["Dispatch_Id"]
    if did not in seen:
        seen.add(did)
        a["n"] += 1
        a["ns"] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
PEAK_HZ = 2.4e9
out = {"formula": "mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel duration x 2.4 GHz)", "kernels": {}}
tot_busy = tot_ns = 0.0
for k, a in sorted(agg.items(), key=lambda kv: -kv[1]["ns"]):
    if a["ns"] <= 0:
        continue
    out["kernels"][k] = {"launches": a["n"], "mfma_busy_cycles_per_launch": a["busy"] / a["n"],
                         "gui_active_cycles_per_launch": a["active"] / a["n"], "us_per_launch": a["ns"] / a["n"] / 1e3,
                         "mfma_busy": a["busy"] / (SIMDS * a["ns"] * 1e-9 * PEAK_HZ),
                         "gui_active_per_ns_raw": a["active"] / a["ns"] if a["ns"] else None}
    tot_busy += a["busy"]
    tot_ns += a["ns"]
out["whole_run_mfma_busy"] = tot_busy / (SIMDS * tot_ns * 1e-9 * PEAK_HZ) if tot_ns else None
json.dump(out, open(sys.argv[2], "w"), indent=1)
for k, v in out["kernels"].items():
    print(f"{100 * v['mfma_busy']:6.1f} %  {v['us_per_launch']:8.1f} us x {v['launches']:4d}  {k[:100]}")
print("whole run:", out["whole_run_mfma_busy"])
