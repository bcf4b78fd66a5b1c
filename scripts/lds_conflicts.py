#!/usr/bin/env python3
"""LDS bank conflicts per kernel from ONE rocprofv3 pass with `--pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE`
(plus --kernel-trace only).  /opt/skills/guides/MI355X_MICROARCH.md: SQ_LDS_BANK_CONFLICT = extra LDS cycles lost to
conflicts, SQ_LDS_IDX_ACTIVE = all LDS-array cycles; reported: conflict cycles / active cycles per kernel.

    python scripts/lds_conflicts.py <counter_collection.csv> <out.json>
"""
import collections
import csv
import json
import sys

agg = collections.defaultdict(lambda: {"n": 0, "conf": 0.0, "active": 0.0, "ns": 0.0})
seen = set()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")
    a = agg[k]
    v = float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_LDS_BANK_CONFLICT":
        a["conf"] += v
    elif r["Counter_Name"] == "SQ_LDS_IDX_ACTIVE":
        a["active"] += v
    if r["Dispatch_Id"] not in seen:
        seen.add(r["Dispatch_Id"])
        a["n"] += 1
        a["ns"] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
out = {"formula": "conflict_frac = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE (cycles summed over the CUs)", "kernels": {}}
for k, a in sorted(agg.items(), key=lambda kv: -kv[1]["ns"]):
    if a["active"] <= 0:
        continue
    out["kernels"][k] = {"launches": a["n"], "us_per_launch": a["ns"] / a["n"] / 1e3,
                         "lds_active_cycles_per_launch": a["active"] / a["n"],
                         "lds_conflict_cycles_per_launch": a["conf"] / a["n"], "conflict_frac": a["conf"] / a["active"]}
json.dump(out, open(sys.argv[2], "w"), indent=1)
for k, v in out["kernels"].items():
    print(f"{100 * v['conflict_frac']:6.2f} % of {v['lds_active_cycles_per_launch']:12.0f} LDS cycles  {v['us_per_launch']:8.1f} us x {v['launches']:4d}  {k[:90]}")
