#!/usr/bin/env python3
"""Per-kernel averages of whatever counters a rocprofv3 --pmc pass collected (one or several counter_collection.csv).
    python scripts/pmc_generic.py <out.json> <counter_collection.csv> [...]  [--match substring]"""
import collections
import csv
import json
import sys

args = [a for a in sys.argv[1:] if not a.startswith("--")]
match = sys.argv[sys.argv.index("--match") + 1] if "--match" in sys.argv else ""
if match in args:
    args.remove(match)
out_path, files = args[0], args[1:]
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, set()]))
dur = collections.defaultdict(lambda: [0.0, set()])
for f in files:
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if match and match not in k:
            continue
        a = agg[k][r["Counter_Name"]]
        a[0] += float(r["Counter_Value"])
        a[1].add((f, r["Dispatch_Id"]))
        d = dur[k]
        if (f, r["Dispatch_Id"]) not in d[1]:
            d[1].add((f, r["Dispatch_Id"]))
            d[0] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
out = {}
for k in sorted(agg, key=lambda k: -dur[k][0]):
    out[k] = {"us_per_launch": dur[k][0] / len(dur[k][1]) / 1e3}
    for c, (v, ids) in sorted(agg[k].items()):
        out[k][c] = v / len(ids)
json.dump(out, open(out_path, "w"), indent=1)
for k, v in out.items():
    print(k[:100])
    print("   ", ", ".join(f"{c}={x:.4g}" for c, x in v.items()))
