#!/usr/bin/env python3
"""Reads a rocprofv3 --kernel-trace CSV of `bench.py --emulate-root-ingress G` and reports, for the streamed tower kernels,
their duration when a k_copy_limited kernel was / was not in flight during the launch, and the copies' own durations.
    python scripts/ingress_overlap.py <kernel_trace.csv>"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
copies = [(s, e) for s, e, n in ks if "k_copy_limited" in n]
tower = [(s, e, n) for s, e, n in ks if "k_ref_block_stream" in n and ", false>" in n and "<1, 64" in n]
if not copies:
    print("no k_copy_limited launches in the trace")
    sys.exit(0)
cd = sorted((e - s) / 1e3 for s, e in copies)
print(f"k_copy_limited: {len(copies)} launches, duration us min/med/max = {cd[0]:.0f}/{cd[len(cd)//2]:.0f}/{cd[-1]:.0f}")
lo, hi = copies[0][0], copies[-1][1]
with_c, without = [], []
for s, e, n in tower:
    ov = sum(max(0, min(e, ce) - max(s, cs)) for cs, ce in copies)
    (with_c if ov > 0.5 * (e - s) else without if ov == 0 else []).append((e - s) / 1e3)
for name, v in (("copy in flight for > half of the launch", with_c), ("no copy in flight", without)):
    if v:
        v.sort()
        print(f"dilation-1 streamed block, {name}: n={len(v)} duration us min/med/max = {v[0]:.0f}/{v[len(v)//2]:.0f}/{v[-1]:.0f}")
