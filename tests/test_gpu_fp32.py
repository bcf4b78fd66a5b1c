"""SN_PREC_FP32 (BASELINE configs[1]) kernels of round 6: the head on the fp32 MFMA (k_head_final_mfma32) against the
per-pixel kernel it replaces and against the CPU oracle; the half-tile work distribution of k_ref_conv_f32 at sizes where
workgroups own whole tiles, single halves and empty right halves.  Needs an MI355X."""
import os
import subprocess
import sys

import numpy as np
import pytest

from hobot_stereonet_amd import api, synth

pytestmark = pytest.mark.gpu
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))

_SCRIPT = """
import sys
sys.path.insert(0, sys.argv[1])
import numpy as np
from hobot_stereonet_amd import api, synth
w, h, d, n = (int(v) for v in sys.argv[4:8])
xs = np.stack([synth.model_input_i8(w, h, d, 70 + i) for i in range(n)])
with api.StereoNetHIP(sys.argv[2], max_batch=n, precision=api.PREC_FP32) as eng:
    disp, raw = eng.infer(xs)
    st = eng.refine_stats()
np.savez(sys.argv[3], disp=disp, raw=raw, residual=st["residual_px"])
"""


@pytest.mark.parametrize("w,h,d,n,multi", [(200, 120, 64, 3, False), (1280, 720, 192, 1, False), (416, 128, 64, 2, True)])
def test_fp32_head_on_the_matrix_core(model_factory, oracle, weights_blob, weights_multi, tmp_path, w, h, d, n, multi):
    """k_head_final_mfma32 == k_head_final up to the fp32 summation order (channels first, then taps), both inside the
    oracle's bound; the refinement statistic is the same sum."""
    model = model_factory(w, h, d, multi)
    script = tmp_path / "run.py"
    script.write_text(_SCRIPT)
    outs = {}
    for tag, e in (("mfma", {}), ("valu", {"SN_HEAD_MFMA32": "0"})):
        out = str(tmp_path / f"{tag}.npz")
        r = subprocess.run([sys.executable, str(script), ROOT, model, out, str(w), str(h), str(d), str(n)],
                           env=dict(os.environ, **e), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs[tag] = np.load(out)
    a, b = outs["mfma"]["disp"].reshape(n, h, w), outs["valu"]["disp"].reshape(n, h, w)
    assert np.abs(a - b).max() < 2e-4, np.abs(a - b).max()
    assert np.abs(outs["mfma"]["raw"].astype(np.int64) - outs["valu"]["raw"]).max() <= 64     # 2e-4 px on the wire scale
    assert abs(float(outs["mfma"]["residual"]) - float(outs["valu"]["residual"])) < 1e-4
    blob = weights_multi if multi else weights_blob
    for i in (0, n - 1):
        odisp, _, _ = oracle.forward(blob, synth.model_input_i8(w, h, d, 70 + i), d)
        assert np.abs(a[i] - odisp).mean() < 2e-4


@pytest.mark.parametrize("h,w,dil", [(8, 32, 1), (8, 96, 1), (24, 160, 2), (720, 1280, 2), (720, 1280, 8), (368, 1248, 4), (100, 100, 1)])
def test_tower_conv_fp32_half_tile_shares(model_factory, oracle, h, w, dil):
    """k_ref_conv_f32 hands every workgroup a contiguous range of HALF tiles: one half per workgroup (tiny maps), ranges
    that start or end inside a tile (1280x720: 7.03 halves per workgroup), right halves outside the image (w % 64 <= 32)."""
    rng = np.random.default_rng(h * 1000 + w + dil)
    x = rng.standard_normal((32, h, w)).astype(np.float32)
    wt = (rng.standard_normal((32, 32, 3, 3)) / 17.0).astype(np.float32)
    b = rng.standard_normal(32).astype(np.float32)
    ref = oracle.conv2d(x, wt, b, 1, dil, dil)
    res = rng.standard_normal((32, h, w)).astype(np.float32)
    v = ref + res
    ref2 = np.where(v > 0, v, v * np.float32(0.2))
    with api.StereoNetHIP(model_factory(96, 64, 48), max_batch=2, precision=api.PREC_FP32) as eng:
        got = eng.dbg_conv2d(x, wt, b, 3, 1, dil, tower32=True)
        got2 = eng.dbg_conv2d(x, wt, b, 3, 1, dil, lrelu=True, residual=res, tower32=True)
    scale = np.abs(ref).max()
    assert np.abs(got - ref).max() / scale < 2e-5
    assert np.abs(got2 - ref2).max() / np.abs(ref2).max() < 2e-5
