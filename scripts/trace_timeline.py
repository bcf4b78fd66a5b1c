#!/usr/bin/env python3
"""Timeline view of a rocprofv3 --kernel-trace CSV: per HW queue, the durations of the tower launches and the gaps
between consecutive launches; plus how much of the wall time has 0 / 1 / 2+ kernels in flight.
    python scripts/trace_timeline.py <kernel_trace.csv> [name-substring]"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
match = sys.argv[2] if len(sys.argv) > 2 else "k_ref_conv_f16_v2"
ks = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r["Queue_Id"]) for r in rows]
ks.sort()
t0 = ks[0][0]
# last third of the run = steady state of the last step
lo = ks[len(ks) * 2 // 3][0]
sel = [k for k in ks if k[0] >= lo]
byq = defaultdict(list)
for s, e, n, q in sel:
    byq[q].append((s, e, n))
for q, lst in sorted(byq.items()):
    tower = [(s, e) for s, e, n in lst if match in n]
    if not tower:
        print(f"queue {q}: {len(lst)} kernels, none match")
        continue
    durs = [(e - s) / 1e3 for s, e in tower]
    gaps = [(lst[i + 1][0] - lst[i][1]) / 1e3 for i in range(len(lst) - 1) if match in lst[i][2] and match in lst[i + 1][2]]
    durs.sort()
    gaps.sort()
    print(f"queue {q}: {len(lst)} kernels, {len(tower)} tower launches; duration us min/med/max = "
          f"{durs[0]:.1f}/{durs[len(durs)//2]:.1f}/{durs[-1]:.1f}; gap to next tower launch on this queue min/med/max = "
          + (f"{gaps[0]:.1f}/{gaps[len(gaps)//2]:.1f}/{gaps[-1]:.1f}" if gaps else "-"))
# concurrency histogram over the selected window
ev = []
for s, e, n, q in sel:
    ev.append((s, 1))
    ev.append((e, -1))
ev.sort()
cur, last, hist = 0, ev[0][0], defaultdict(int)
for t, d in ev:
    hist[min(cur, 3)] += t - last
    last = t
    cur += d
tot = sum(hist.values())
print("kernels in flight: " + ", ".join(f"{k}{'+' if k == 3 else ''}: {100.0 * v / tot:.1f} %" for k, v in sorted(hist.items())),
      f"(window {tot / 1e6:.2f} ms, {len(sel)} kernels)")
# per-kernel totals in the window
agg = defaultdict(lambda: [0, 0.0])
for s, e, n, q in sel:
    key = n.split("(")[0][:90]
    agg[key][0] += 1
    agg[key][1] += (e - s) / 1e3
for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:14]:
    print(f"  {t:9.1f} us  {c:5d} x {t / c:7.1f}  {k}")
