"""The tail form of the fp16 tower — the last residual block and the refinement head in ONE launch
(k_ref_block_stream_f16<..., HEAD = true>) — against the two launches it replaces (streamed block +
k_head_final_f16) BIT FOR BIT, at the sizes the pipeline runs it (a mean-EPE bound would pass a wrong
column at a strip seam: 60-column strips, 22 per 1280-wide row, low-resolution windows staged by 4-byte LDS-DMA), and
against the CPU oracle's arithmetic.  The network behind DnnNode::Run (stereonet_infer/src/stereonet_node.cpp:812).
Needs an MI355X."""
import os
import subprocess
import sys

import numpy as np
import pytest

from hobot_stereonet_amd import api, spec, synth

pytestmark = pytest.mark.gpu
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def q16(a):
    return np.asarray(a, np.float32).astype(np.float16).astype(np.float32)


@pytest.fixture(scope="module")
def eng16(model_factory):
    eng = api.StereoNetHIP(model_factory(96, 64, 48), max_batch=2, precision=api.PREC_F16)
    yield eng
    eng.close()


def _operands(seed, n, hk, wk, ups):
    rng = np.random.default_rng(seed)
    x = q16(rng.standard_normal((n, 32, hk, wk)))
    w1 = q16(rng.standard_normal((32, 32, 3, 3)) / 17.0)
    w2 = q16(rng.standard_normal((32, 32, 3, 3)) / 17.0)
    b1 = rng.standard_normal(32).astype(np.float32)
    b2 = rng.standard_normal(32).astype(np.float32)
    hw = (rng.standard_normal((32, 9)) * 0.004 / np.sqrt(288.0)).astype(np.float32)
    hb = float(rng.standard_normal() * 0.0005)
    # a disparity-like map: smooth ramp + noise, some zeros (the relu at the output must clip there)
    low = (rng.random((n, hk // ups, wk // ups)) * 11.0).astype(np.float32)
    low[:, : max(1, hk // ups // 4)] = 0.0
    return x, w1, b1, w2, b2, hw, hb, low


# (hk, wk, h_out, w_out, ups, n): level-0 geometries (x16 from the soft-argmin map; the activation is the padded size, the
# maps the image size) and coarse levels of the hierarchical model (x2 from the level below; whole padded level)
TAIL_CASES = [
    (720, 1280, 720, 1280, 16, 1),      # the metric's shape: 22 strips of 60 columns, 20 of them interior
    (384, 1248, 375, 1242, 16, 2),      # KITTI, two images: output smaller than the activation on both axes
    (112, 144, 100, 129, 16, 1),        # width not a multiple of 60 / 64 / 32
    (48, 256, 37, 250, 16, 2),
    (32, 80, 31, 67, 16, 1),            # narrower than two strips
    (16, 64, 16, 61, 16, 1),            # ONE strip and one column
    (192, 624, 192, 624, 2, 1),         # level 1 of 1242x375: x2 windows
    (96, 312, 96, 312, 2, 3),           # level 2, three images
    (48, 156, 48, 156, 2, 1),           # level 3
    (360, 640, 360, 640, 2, 1),         # level 1 of 1280x720
]


@pytest.mark.parametrize("hk,wk,ho,wo,ups,n", TAIL_CASES)
def test_tail_form_equals_block_plus_head_bit_for_bit(eng16, hk, wk, ho, wo, ups, n):
    ops = _operands(hk * 13 + wk + ups, n, hk, wk, ups)
    dnorm = 192.0 if ups == 16 else 96.0
    d0, r0 = eng16.dbg_ref_tail_f16(*ops, ups, dnorm, ho, wo, form=0)
    d1, r1 = eng16.dbg_ref_tail_f16(*ops, ups, dnorm, ho, wo, form=1)
    assert np.isfinite(d0).all() and np.isfinite(d1).all()          # every pixel written (the hook pre-fills with NaN / -1)
    assert (r0 >= 0).all() and (r1 >= 0).all()
    bad = np.argwhere(d0 != d1)
    assert bad.size == 0, f"{len(bad)} pixels differ, first at (image, row, column) {bad[0]}: {d0[tuple(bad[0])]} vs {d1[tuple(bad[0])]}"
    assert np.array_equal(r0, r1)
    inv_q = np.float32(1.0 / (192.0 * float(np.float32(spec.OUT_SCALE))))
    assert np.array_equal(r1, np.rint(d1 * inv_q).astype(np.int32))
    # the map is not trivial and the relu clips somewhere (the one-strip case starts from an all-zero 1 x 4 map: only D * r)
    assert d1.std() > (0.5 if hk > 16 else 0.1) and (d1 == 0).any()


@pytest.mark.parametrize("hk,wk,ho,wo,ups", [(112, 144, 100, 129, 16), (48, 156, 48, 156, 2)])
def test_tail_form_vs_oracle_arithmetic(eng16, oracle, hk, wk, ho, wo, ups):
    """Not only equal to each other: both forms compute block + head.  Reference = oracle convs on fp16-rounded operands
    (t and y rounded to fp16 as the tensors hold them), head in fp32, bilinear upsample as the oracle's refinement does it."""
    x, w1, b1, w2, b2, hw, hb, low = _operands(5, 1, hk, wk, ups)
    dnorm = 192.0 if ups == 16 else 96.0
    lr = lambda v: np.where(v > 0, v, v * np.float32(0.2))
    t = q16(lr(oracle.conv2d(x[0], w1, b1, 1, 1, 1)))
    y = q16(lr(x[0] + oracle.conv2d(t, w2, b2, 1, 1, 1)))
    r = oracle.conv2d(y, hw.reshape(1, 32, 3, 3), np.array([hb], np.float32), 1, 1, 1)[0]
    up = oracle.upsample_bilinear(low[0], ups, float(ups))          # values x factor
    ref = np.maximum(up[:hk, :wk] + np.float32(dnorm) * r, 0)[:ho, :wo]
    got, _ = eng16.dbg_ref_tail_f16(x, w1, b1, w2, b2, hw, hb, low, ups, dnorm, ho, wo, form=1)
    err = np.abs(got[0] - ref)
    print(f"tail form vs oracle arithmetic {hk}x{wk} x{ups}: mean {err.mean():.2e} max {err.max():.2e}")
    assert err.mean() < 2e-4 and err.max() < 5e-3


_ENV_SCRIPT = r"""
import sys, numpy as np
sys.path.insert(0, sys.argv[1])
from hobot_stereonet_amd import api, synth
w, h, d, n, multi = int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6]), int(sys.argv[7]), sys.argv[8] == "1"
xs = np.stack([synth.model_input_i8(w, h, d, 90 + i) for i in range(n)])
with api.StereoNetHIP(sys.argv[2], max_batch=n, precision=api.PREC_F16) as eng:
    disp, raw = eng.infer(xs)
    lv = [eng.dbg_read(f"level{k}") for k in range(1, eng.refine_levels)]
np.savez(sys.argv[3], disp=disp, raw=raw, **{f"level{k + 1}": v for k, v in enumerate(lv)})
"""


@pytest.mark.parametrize("w,h,d,n,multi", [(1280, 720, 192, 5, False),       # default refine_chunk 4: a ragged chunk of one pair
                                           (1242, 375, 256, 3, True)])      # four tail launches per level set, x16 and x2 windows
def test_tail_fuse_switch_is_bit_identical_at_size(model_factory, tmp_path, w, h, d, n, multi):
    """SN_TAIL_FUSE=0 (block + k_head_final_f16) against the default (tail form) through the whole engine at the sizes the
    bench runs: every output pixel, the wire maps, and for the hierarchical model the map of every coarse level."""
    model = model_factory(w, h, d, multi=multi)
    script = tmp_path / "run.py"
    script.write_text(_ENV_SCRIPT)
    outs = {}
    for tag, e in (("default", {}), ("two_launches", {"SN_TAIL_FUSE": "0"})):
        out = str(tmp_path / f"{tag}.npz")
        r = subprocess.run([sys.executable, str(script), ROOT, model, out, str(w), str(h), str(d), str(n), "1" if multi else "0"],
                           env=dict(os.environ, **e), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs[tag] = np.load(out)
    a, b = outs["default"], outs["two_launches"]
    assert sorted(a.files) == sorted(b.files) and len(a.files) == (5 if multi else 2)
    for k in a.files:
        assert np.array_equal(a[k], b[k]), k
    assert a["disp"].shape == (n, h, w) and np.isfinite(a["disp"]).all()
