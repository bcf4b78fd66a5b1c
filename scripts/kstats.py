#!/usr/bin/env python3
"""Pretty-prints a rocprofv3 *_kernel_stats.csv (per-kernel calls / average / total / share)."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(int(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot / 1e6:.2f} ms")
for r in rows[: int(sys.argv[2]) if len(sys.argv) > 2 else 18]:
    name = r["Name"].replace("HIP_vector_type<unsigned int, 4u>", "uint4")
    print(f"{name[:88]:88s} n={r['Calls']:>4s} avg_us={float(r['AverageNs']) / 1e3:8.1f} "
          f"tot_ms={int(r['TotalDurationNs']) / 1e6:7.2f} {float(r['Percentage']):5.1f}%")
