#!/usr/bin/env python3
"""Reads a rocprofv3 *_kernel_trace.csv and reports how much the two streams of the piece pipeline
actually overlap: per queue the summed kernel time, the union of all kernel intervals, and the time
during which kernels of two different queues were in flight together.

    python scripts/overlap.py <kernel_trace.csv> [skip_fraction]

skip_fraction (default 0.5) drops the leading part of the trace (engine creation, warm-up).
"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
ev = []
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    ev.append((s, e, r.get("Queue_Id", "?"), r["Kernel_Name"]))
ev.sort()
t_lo, t_hi = ev[0][0], max(e for _, e, _, _ in ev)
cut = t_lo + (t_hi - t_lo) * skip
ev = [x for x in ev if x[0] >= cut]
span = max(e for _, e, _, _ in ev) - ev[0][0]

per_q = defaultdict(int)
per_k = defaultdict(lambda: [0, 0])
for s, e, q, k in ev:
    per_q[q] += e - s
    kk = k.split("(")[0][-60:]
    per_k[(q, kk)][0] += 1
    per_k[(q, kk)][1] += e - s

# sweep: time with >=1 kernel, time with kernels from >=2 distinct queues
pts = []
for s, e, q, _ in ev:
    pts.append((s, 1, q))
    pts.append((e, -1, q))
pts.sort()
live = defaultdict(int)
busy = both = 0
prev = pts[0][0]
for t, d, q in pts:
    nq = sum(1 for v in live.values() if v > 0)
    if nq >= 1:
        busy += t - prev
    if nq >= 2:
        both += t - prev
    live[q] += d
    prev = t

print(f"window {span / 1e6:.2f} ms, kernels {len(ev)}")
print(f"busy (>=1 kernel in flight) {busy / 1e6:.2f} ms = {100.0 * busy / span:.1f}% of window")
print(f"two queues in flight together {both / 1e6:.2f} ms = {100.0 * both / span:.1f}% of window")
for q, t in sorted(per_q.items()):
    print(f"queue {q}: summed kernel time {t / 1e6:.2f} ms")
print("per kernel (queue, name): calls, avg us")
for (q, k), (n, t) in sorted(per_k.items(), key=lambda kv: -kv[1][1])[:24]:
    print(f"  q{q} {k:60s} n={n:5d} avg={t / n / 1e3:8.1f} us tot={t / 1e6:7.2f} ms")
