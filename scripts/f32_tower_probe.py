"""One fp32 tower layer (k_ref_conv_f32) at 1280x720 through sn_dbg_conv2d, a few launches: run under rocprofv3 --stats.
    SN_F32_GRID=n   workgroups per launch (default: two per CU)"""
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hobot_stereonet_amd import api, weights  # noqa: E402

dil = int(sys.argv[1]) if len(sys.argv) > 1 else 1
h, w = 720, 1280
rng = np.random.default_rng(0)
x = rng.standard_normal((32, h, w)).astype(np.float32)
wt = (rng.standard_normal((32, 32, 3, 3)) / 17.0).astype(np.float32)
b = rng.standard_normal(32).astype(np.float32)
with tempfile.TemporaryDirectory() as td:
    path = os.path.join(td, "m.snw")
    weights.save_snw(path, weights.synthetic(0), 96, 64, 48)
    with api.StereoNetHIP(path, precision=api.PREC_FP32) as eng:
        for _ in range(6):
            eng.dbg_conv2d(x, wt, b, 3, 1, dil, tower32=True)
