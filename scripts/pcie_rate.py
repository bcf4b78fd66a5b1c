#!/usr/bin/env python3
"""PCIe-inclusive rate of the boundary when it is handed HOST buffers (what DnnNode::Run does in the node):
(a) synchronous sn_infer_batch on pageable numpy arrays, (b) the async sn_submit/sn_wait path with task_num
requests in flight through pinned staging.  Reported in DESIGN.md §6; never bench.py's `value`."""
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from hobot_stereonet_amd import api, synth, weights  # noqa: E402

W, H, D = 1280, 720, 192
tmp = tempfile.mkdtemp()
model = os.path.join(tmp, "m.snw")
weights.save_snw(model, weights.synthetic(0), W, H, D)
xs = np.stack([synth.model_input_i8(W, H, D, i) for i in range(4)])
xs = np.concatenate([xs] * 4)          # 16 pairs
with api.StereoNetHIP(model, max_batch=16, precision=api.PREC_F16, task_num=4) as eng:
    eng.infer(xs)                       # warm-up
    t0 = time.perf_counter()
    for _ in range(3):
        eng.infer(xs, want_disp=False)
    dt = time.perf_counter() - t0
    print(f"sync host batch-16 (pageable, int32 out only): {3 * 16 / dt:.1f} pairs/s")
    outs = [np.empty((H, W), np.int32) for _ in range(4)]
    n = 64
    tickets = []
    t0 = time.perf_counter()
    for i in range(n):
        if len(tickets) == 4:
            eng.wait(tickets.pop(0))
        tickets.append(eng.submit(xs[i % 16], outs[i % 4], None))
    for t in tickets:
        eng.wait(t)
    dt = time.perf_counter() - t0
    print(f"async submit/wait, 4 in flight (pinned staging): {n / dt:.1f} pairs/s, {dt / n * 1e3:.2f} ms/frame")
