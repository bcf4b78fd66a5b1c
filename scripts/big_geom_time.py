#!/usr/bin/env python3
"""Throughput of a geometry outside bench.py's configs (default 2048x1088 D=256, batch 8), inputs resident in HBM:
    python scripts/big_geom_time.py [W H D batch]"""
import os, sys, tempfile, time
import numpy as np
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
from hobot_stereonet_amd import api, synth, weights
w, h, d, n = (int(v) for v in (sys.argv[1:5] if len(sys.argv) >= 5 else (2048, 1088, 256, 8)))
m = os.path.join(tempfile.mkdtemp(), "m.snw")
weights.save_snw(m, weights.synthetic(0), w, h, d)
base = synth.model_input_i8(w, h, d, 5)
x = torch.from_numpy(np.stack([np.roll(base, 16 * i, axis=2) for i in range(n)])).cuda()
raw = torch.empty((n, h, w), dtype=torch.int32, device="cuda")
disp = torch.empty((n, h, w), dtype=torch.float32, device="cuda")
with api.StereoNetHIP(m, max_batch=n, precision=api.PREC_F16) as eng:
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(3):
        eng.infer_device(n, x.data_ptr(), raw.data_ptr(), disp.data_ptr(), st)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    steps = 20
    for _ in range(steps):
        eng.infer_device(n, x.data_ptr(), raw.data_ptr(), disp.data_ptr(), st)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{w}x{h} D={d} batch {n} refine_chunk {eng.refine_chunk}: {n * steps / dt:.1f} pairs/s  (SN_REV={os.environ.get('SN_REV', 'default')})")
