#!/usr/bin/env python3
"""Per-kernel averages of the counters in one or more rocprofv3 *_counter_collection.csv files.
    python scripts/pmc_kernel.py <match> <csv> [<csv> ...]"""
import collections
import csv
import sys

match = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for path in sys.argv[2:]:
    for r in csv.DictReader(open(path)):
        if match not in r["Kernel_Name"]:
            continue
        k = r["Kernel_Name"].split("(")[0][-70:]
        a = agg[k][r["Counter_Name"]]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
for k, cs in agg.items():
    print(k)
    for c, (n, v) in sorted(cs.items()):
        print(f"   {c:32s} n={n:4d} avg={v / n:16.1f}")
