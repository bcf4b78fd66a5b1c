"""The row-streaming fused residual block on split operands (k_ref_block_stream_x3, SN_PREC_F16X3 and the arithmetic an
SN_PREC_AUTO handle falls back to): bit-identical to the two k_ref_conv_f16x3 launches it replaces, inside the fp32 oracle's
tolerance, borders intact (the hook checks them).  Needs an MI355X."""
import numpy as np
import pytest

from hobot_stereonet_amd import api

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng(model_factory):
    e = api.StereoNetHIP(model_factory(96, 64, 48), max_batch=1, precision=api.PREC_F16X3)
    yield e
    e.close()


def _block(oracle, x, w1, b1, w2, b2, dil):
    t = oracle.conv2d(x, w1, b1, 1, dil, dil)
    t = np.where(t > 0, t, t * np.float32(0.2))
    y = oracle.conv2d(t, w2, b2, 1, dil, dil) + x
    return np.where(y > 0, y, y * np.float32(0.2))


@pytest.mark.parametrize("h,w,dil", [(16, 64, 1), (45, 80, 1), (64, 128, 1), (33, 70, 2), (90, 160, 2), (72, 200, 4), (130, 300, 4),
                                     (20, 20, 4), (7, 130, 1), (375, 1242, 1), (720, 1280, 2), (720, 1280, 4), (130, 304, 8), (20, 20, 8),
                                     (64, 96, 8), (720, 1280, 8)])
def test_streamed_split_block_is_the_two_launches(eng, oracle, h, w, dil):
    rng = np.random.default_rng(h * 1000 + w + dil)
    x = rng.standard_normal((32, h, w)).astype(np.float32)
    w1 = (rng.standard_normal((32, 32, 3, 3)) / 17.0).astype(np.float32)
    w2 = (rng.standard_normal((32, 32, 3, 3)) / 17.0).astype(np.float32)
    b1 = rng.standard_normal(32).astype(np.float32)
    b2 = rng.standard_normal(32).astype(np.float32)
    two = eng.dbg_ref_block_f16x3(x, w1, b1, w2, b2, dil, streamed=False)
    one = eng.dbg_ref_block_f16x3(x, w1, b1, w2, b2, dil, streamed=True)
    assert np.array_equal(one, two), float(np.abs(one - two).max())
    if h * w <= 130 * 300:
        ref = _block(oracle, x, w1, b1, w2, b2, dil)
        assert np.abs(one - ref).max() / np.abs(ref).max() < 2e-5


_SCRIPT = """
import sys
sys.path.insert(0, sys.argv[1])
import numpy as np
from hobot_stereonet_amd import api, synth
w, h, d, n = (int(v) for v in sys.argv[4:8])
xs = np.stack([synth.model_input_i8(w, h, d, 90 + i) for i in range(n)])
with api.StereoNetHIP(sys.argv[2], max_batch=n, precision=api.PREC_F16X3) as eng:
    disp, raw = eng.infer(xs)
np.savez(sys.argv[3], disp=disp, raw=raw)
"""


@pytest.mark.parametrize("w,h,d,n,multi", [(200, 120, 64, 3, False), (416, 128, 64, 2, True), (1280, 720, 192, 1, False)])
def test_split_mode_end_to_end_is_unchanged_by_the_streamed_blocks(model_factory, oracle, weights_blob, weights_multi, tmp_path,
                                                                  w, h, d, n, multi):
    """SN_PREC_F16X3 through the pipeline with the streamed blocks (default) and with SN_X3_STREAM=0 (two launches per block):
    the same maps bit for bit, single-scale and hierarchical, and inside the split mode's bound against the oracle."""
    import os
    import subprocess
    import sys
    from hobot_stereonet_amd import synth
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    model = model_factory(w, h, d, multi)
    script = tmp_path / "run.py"
    script.write_text(_SCRIPT)
    outs = {}
    for tag, e in (("stream", {}), ("layers", {"SN_X3_STREAM": "0"}), ("nwr2", {"SN_X3_NWR": "2"})):
        out = str(tmp_path / f"{tag}.npz")
        r = subprocess.run([sys.executable, str(script), root, model, out, str(w), str(h), str(d), str(n)],
                           env=dict(os.environ, **e), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs[tag] = np.load(out)
    assert np.array_equal(outs["stream"]["raw"], outs["layers"]["raw"])
    assert np.array_equal(outs["stream"]["disp"], outs["layers"]["disp"])
    assert np.array_equal(outs["stream"]["raw"], outs["nwr2"]["raw"])          # one wave per SIMD (SN_X3_NWR=2): the same bits
    blob = weights_multi if multi else weights_blob
    odisp, _, _ = oracle.forward(blob, synth.model_input_i8(w, h, d, 90), d)
    assert np.abs(outs["stream"]["disp"].reshape(n, h, w)[0] - odisp).mean() < 2e-4
