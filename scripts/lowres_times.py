#!/usr/bin/env python3
"""Per-launch times of the low-resolution branch from a rocprofv3 --kernel-trace CSV (last piece of the run)."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
seq = [(r["Kernel_Name"].split("(")[0].replace("void sn::", ""), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3,
        r["Grid_Size_X"]) for r in rows]
idx = [i for i, s in enumerate(seq) if s[0].startswith("k_down0")]
i0 = idx[-1]
tot = 0.0
for s in seq[i0:i0 + 24]:
    if s[0].startswith(("k_refin", "k_ref_conv", "k_head_final")):
        break
    tot += s[1]
    print(f"{s[1]:8.1f} us grid {s[2]:>8}  {s[0][:90]}")
print(f"total {tot:.1f} us")
