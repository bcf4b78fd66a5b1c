#!/usr/bin/env python3
"""Per-position duration of the refinement tower's launches, from a rocprofv3 --kernel-trace CSV: the tower of one
chunk is a fixed sequence of launches (ref.in, six streamed residual blocks, the head — or, with SN_FUSE=0, 6 x 2 convs
with the last one fused with the head); this prints,
for each position of the sequence, the median / min / max duration over all chunks of the run.
    python scripts/tower_sequence.py <kernel_trace.csv> [levels]
levels = 4 for a hierarchical model: the towers of levels 3, 2, 1, 0 follow each other per chunk; the summary is then
printed per level (sequence index modulo levels)."""
import csv
import re
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
seq = []
for s, e, n in ks:
    if "k_ref_conv_f16_v2" in n or "k_refin_f16" in n or "k_ref_block_stream_f16" in n or "k_head_final_f16" in n:
        m = re.search(r"k_ref_conv_f16_v2<(\d+), (\d+), (true|false)", n)
        b = re.search(r"k_ref_block_stream_f16<(\d+), (\d+), (\d+)", n)
        if "k_refin" in n:
            tag = "refin"
        elif "head" in n:
            tag = "head"
        elif b:
            tag = f"block dil{b.group(1)} {b.group(2)}x{b.group(3)}"
        else:
            tag = f"dil{m.group(1)} tw{m.group(2)} res{int(m.group(3) == 'true')}"
        seq.append((tag, (e - s) / 1e3))
# split into chunks at every refin launch
chunks, cur = [], []
for tag, d in seq:
    if tag == "refin":
        if cur:
            chunks.append(cur)
        cur = []
    cur.append((tag, d))
if cur:
    chunks.append(cur)
full = max(len(c) for c in chunks)
levels = int(sys.argv[2]) if len(sys.argv) > 2 else 1
chunks = chunks[len(chunks) // 3 // levels * levels:]
for lv in range(levels):
    sel = [c for i, c in enumerate(chunks) if i % levels == lv and len(c) == full]
    if not sel:
        continue
    print(f"{len(sel)} towers of {full} launches" + (f" — level {levels - 1 - lv}" if levels > 1 else ""))
    tot = 0.0
    for i in range(full):
        ds = sorted(c[i][1] for c in sel)
        tot += ds[len(ds) // 2]
        print(f"  {i:2d} {sel[0][i][0]:18s} med {ds[len(ds)//2]:7.1f} us   min {ds[0]:7.1f}   max {ds[-1]:7.1f}")
    print(f"  sum of medians {tot:.1f} us")
