import os, sys, numpy as np, tempfile
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "oracle"))
import oracle_py
from hobot_stereonet_amd import api, synth, weights
td = tempfile.mkdtemp(); p = os.path.join(td, 'm.snw'); weights.save_snw(p, weights.synthetic(0), 96, 64, 48)
h, w, dil = 16, 112, 1
rng = np.random.default_rng(1)
x = rng.standard_normal((32, h, w)).astype(np.float32)
wt = (rng.standard_normal((32, 32, 3, 3)) / 17.0).astype(np.float32)
b = rng.standard_normal(32).astype(np.float32)
ref = oracle_py.conv2d(x, wt, b, 1, dil, dil)
with api.StereoNetHIP(p, precision=api.PREC_FP32) as e:
    got = e.dbg_conv2d(x, wt, b, 3, 1, dil, tower32=True)
err = np.abs(got - ref)
print("cols with error:", np.where(err.max((0, 1)) > 1e-3)[0])
print("rows with error:", np.where(err.max((0, 2)) > 1e-3)[0])
print("chans with error:", np.where(err.max((1, 2)) > 1e-3)[0])
c = np.where(err.max((1, 2)) > 1e-3)[0]
if len(c):
    c0 = c[0]; print("channel", c0, "row 0 got", got[c0, 0, 92:112], "\nref", ref[c0, 0, 92:112])
