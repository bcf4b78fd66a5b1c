"""EPE <= 1e-3 px (north_star) as a property of the kernels, not of one weight file: every other parity test, the smoke
and the bench run weights.synthetic(0).  Here: other weight seeds at the two shapes the metric and the stream config are
quoted on, the bound per precision mode, and the documented envelope of the fp16 tower (profiles/r05_epe_sensitivity.txt,
scripts/epe_sensitivity.py): its error grows with the refinement residual D * r, so a head gain beyond the envelope must be
run in SN_PREC_F16X3.  HIP path through the C ABI vs the CPU oracle (the network behind DnnNode::Run,
stereonet_infer/src/stereonet_node.cpp:812; the model file is opaque to the reference, :131-136).  Needs an MI355X."""
import numpy as np
import pytest

from hobot_stereonet_amd import api, spec, synth, weights

pytestmark = pytest.mark.gpu
F16_TOL, X3_TOL = 1e-3, 2e-4
SHAPES = {"c2_single": (1280, 720, 192, 1), "c5_multi": (1242, 375, 256, spec.MULTI_LEVELS)}


def _model(tmp_path, blob, w, h, d):
    p = str(tmp_path / "m.snw")
    weights.save_snw(p, blob, w, h, d)
    return p


def _epe(path, prec, x, od):
    with api.StereoNetHIP(path, precision=prec) as eng:
        disp, raw = eng.infer(x)
    err = np.abs(disp - od)
    assert np.isfinite(disp).all() and raw.min() >= 0
    return float(err.mean()), float(err.max())


@pytest.mark.parametrize("seed", [1, 2, 3, 5, 6, 7])      # 6 and 7: the two worst draws of profiles/r05_epe_sensitivity_*.txt
@pytest.mark.parametrize("shape", sorted(SHAPES))
def test_other_weight_seeds_stay_inside_the_bounds(oracle, tmp_path, shape, seed):
    w, h, d, levels = SHAPES[shape]
    blob = weights.synthetic(seed, levels)
    x = synth.model_input_i8(w, h, d, 500 + seed)
    od = oracle.forward(blob, x, d)[0]
    path = _model(tmp_path, blob, w, h, d)
    e16, m16 = _epe(path, api.PREC_F16, x, od)
    ex3, mx3 = _epe(path, api.PREC_F16X3, x, od)
    print(f"{shape} seed {seed}: F16 EPE {e16:.3e} (max {m16:.2e}), F16X3 EPE {ex3:.3e} (max {mx3:.2e})")
    assert e16 < F16_TOL, e16
    assert ex3 < X3_TOL, ex3


@pytest.mark.parametrize("shape", sorted(SHAPES))
def test_margin_of_the_default_weights(oracle, tmp_path, shape):
    """The seed-0 file every other test uses must keep a margin to the bound (later precision trade-offs must not
    silently exhaust the 1e-3 px budget): F16 below 0.8e-3 at both shapes."""
    w, h, d, levels = SHAPES[shape]
    blob = weights.synthetic(0, levels)
    x = synth.model_input_i8(w, h, d, 22)
    od = oracle.forward(blob, x, d)[0]
    e16, _ = _epe(_model(tmp_path, blob, w, h, d), api.PREC_F16, x, od)
    print(f"{shape} seed 0: F16 EPE {e16:.3e}")
    assert e16 < 0.8 * F16_TOL, e16


@pytest.mark.parametrize("shape", sorted(SHAPES))
def test_large_residuals_need_and_get_the_split_mode(oracle, tmp_path, shape):
    """Head gain 8 (the refinement moves the map by several pixels on average): outside the fp16 tower's envelope by
    construction — its error is 1.8e-4 .. 8.3e-4 px per pixel of residual — and inside SN_PREC_F16X3's and SN_PREC_FP32's.
    A forced SN_PREC_F16 is over the bound there (that is why it is not the default); the default, SN_PREC_AUTO, must notice
    and deliver the split mode's maps (tests/test_gpu_auto.py runs the whole seed x gain grid)."""
    w, h, d, levels = SHAPES[shape]
    blob = weights.synthetic(1, levels, head_gain=8.0)
    x = synth.model_input_i8(w, h, d, 501)
    od, _, olow = oracle.forward(blob, x, d)
    refine_px = float(np.abs(od - oracle.upsample_bilinear(olow, 16, 16.0)[:h, :w]).mean())
    path = _model(tmp_path, blob, w, h, d)
    ex3, _ = _epe(path, api.PREC_F16X3, x, od)
    e32, _ = _epe(path, api.PREC_FP32, x, od)
    e16, _ = _epe(path, api.PREC_F16, x, od)
    print(f"{shape} head gain 8: refinement {refine_px:.2f} px mean; EPE F16 {e16:.3e}, F16X3 {ex3:.3e}, FP32 {e32:.3e}")
    with api.StereoNetHIP(path) as eng:               # default precision
        da, _ = eng.infer(x)
        st = eng.refine_stats()
    ea = float(np.abs(da - od).mean())
    print(f"{shape} head gain 8: default precision ran {st['precision_last']} (residual statistic {st['residual_px']:.2f} px, "
          f"limit {st['limit_px']:.2f} px), EPE {ea:.3e}")
    assert refine_px > 2.0
    assert ex3 < X3_TOL and e32 < X3_TOL
    assert e16 > F16_TOL                               # the fp16 tower alone does NOT hold the bound here
    assert st["precision_last"] == "f16x3" and ea < X3_TOL


@pytest.mark.parametrize("shape,seed", [("c2_single", 6), ("c5_multi", 3)])
def test_sum_preserving_weight_rounding_removes_the_offset(oracle, tmp_path, monkeypatch, shape, seed):
    """The two weight draws on which round-to-nearest fp16 weights cost the most (profiles/r05_epe_sensitivity.txt: 7.8e-4
    px at 1280x720 for seed 6, 1.12e-3 px — over the bound — for the hierarchical model at 1242x375 with seed 3): the
    library's sum-preserving rounding of every 3x3 kernel (stereonet_hip.hip round_kernel_sum_preserving) must take out the
    coherent offset those errors add up to, measured against the oracle on the same input; SN_W_ROUND=rne is the A/B."""
    w, h, d, levels = SHAPES[shape]
    blob = weights.synthetic(seed, levels)
    x = synth.model_input_i8(w, h, d, 500 + seed)
    od = oracle.forward(blob, x, d)[0]
    path = _model(tmp_path, blob, w, h, d)
    res = {}
    for mode in ("rne", "sum"):
        if mode == "rne":
            monkeypatch.setenv("SN_W_ROUND", "rne")
        else:
            monkeypatch.delenv("SN_W_ROUND", raising=False)
        with api.StereoNetHIP(path, precision=api.PREC_F16) as eng:
            disp, _ = eng.infer(x)
        res[mode] = (float(np.abs(disp - od).mean()), float((disp - od).mean()))
    print(f"{shape} seed {seed}: EPE / signed mean error, rne {res['rne'][0]:.3e} / {res['rne'][1]:+.3e}, "
          f"sum-preserving {res['sum'][0]:.3e} / {res['sum'][1]:+.3e}")
    assert res["sum"][0] < F16_TOL
    assert res["sum"][0] < 0.85 * res["rne"][0]
    assert abs(res["sum"][1]) < 0.5 * abs(res["rne"][1])
