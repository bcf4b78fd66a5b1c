"""Hierarchical ("multi") refinement — SURVEY.md appendix A: the refinement tower with separate weights at 1/8, 1/4, 1/2
and full resolution, x2 bilinear between the levels — HIP path vs the CPU oracle through the C ABI.  The model file
says which refinement it holds (weights.py, header word 72); nothing else changes for the caller.  Needs an MI355X."""
import numpy as np
import pytest

from hobot_stereonet_amd import api, spec, synth

pytestmark = pytest.mark.gpu
EPE_TOL = 1e-3      # px, BASELINE.json north_star
CASES = [("c96x64_d48", 96, 64, 48, 3), ("c160x96_d96", 160, 96, 96, 4), ("c100x52_d32", 100, 52, 32, 5)]
# (mean, max) bounds in level pixels for the maps of the coarse levels and the final map
TOLS = {api.PREC_FP32: (1e-4, 2e-3), api.PREC_F16X3: (1e-4, 2e-3), api.PREC_F16: (EPE_TOL, 20 * EPE_TOL)}


@pytest.mark.parametrize("prec", [api.PREC_FP32, api.PREC_F16X3, api.PREC_F16])
@pytest.mark.parametrize("name,w,h,d,seed", CASES)
def test_multi_small_vs_oracle_and_golden(model_factory, oracle, golden_multi, weights_multi, name, w, h, d, seed, prec):
    x = synth.model_input_i8(w, h, d, seed)
    hp, wp = spec.ceil16(h), spec.ceil16(w)
    with api.StereoNetHIP(model_factory(w, h, d, multi=True), precision=prec) as eng:
        assert eng.refine_levels == spec.MULTI_LEVELS
        assert abs(eng.info.flops_per_pair - spec.flops_per_pair(w, h, d, levels=spec.MULTI_LEVELS)) < 1.0
        disp, raw = eng.infer(x)
        maps = [eng.dbg_read(f"level{k}").reshape(hp >> k, wp >> k) for k in range(1, spec.MULTI_LEVELS)]
    odisp, oraw, olow, omaps = oracle.forward_levels(weights_multi, x, d)
    mean_tol, max_tol = TOLS[prec]
    for k, (m, om) in enumerate(zip(maps, omaps), start=1):
        err = np.abs(m - om)
        print(f"{name} prec={prec} level {k}: mean {err.mean():.2e} max {err.max():.2e}")
        assert err.mean() < mean_tol and err.max() < max_tol, (k, err.mean(), err.max())
    err = np.abs(disp - odisp)
    print(f"{name} prec={prec} final: EPE {err.mean():.2e} max {err.max():.2e}")
    assert err.mean() < mean_tol and err.max() < max_tol
    assert np.abs(disp - golden_multi[name + ".disp"]).mean() < EPE_TOL
    inv_q = np.float32(1.0 / (192.0 * float(np.float32(spec.OUT_SCALE))))
    assert (raw == np.rint(disp * inv_q).astype(np.int32)).all() and raw.min() >= 0
    assert np.abs(raw.astype(np.int64) - oraw).max() <= max(1, int(np.ceil(err.max() * float(inv_q))) + 1)


def test_multi_differs_from_single_and_shares_the_lowres_branch(model_factory, oracle, weights_multi):
    """A multi file starts with a complete single file: same low-resolution branch, different refinement."""
    w, h, d = 160, 96, 96
    x = synth.model_input_i8(w, h, d, 4)
    with api.StereoNetHIP(model_factory(w, h, d, multi=True), precision=api.PREC_FP32) as em, \
            api.StereoNetHIP(model_factory(w, h, d), precision=api.PREC_FP32) as es:
        dm, _ = em.infer(x)
        lm = em.dbg_read("disp_low")
        ds, _ = es.infer(x)
        ls = es.dbg_read("disp_low")
        assert es.refine_levels == 1
        with pytest.raises(Exception):
            es.dbg_read("level1")
    assert np.array_equal(lm, ls)
    assert np.abs(dm - ds).mean() > 0.05


@pytest.mark.parametrize("prec", [api.PREC_FP32, api.PREC_F16])
def test_multi_kitti_c5_1242x375_d256(model_factory, oracle, weights_multi, prec):
    """BASELINE.json configs[4] / SURVEY.md §8(d) C5: KITTI 1242x375, D=256, hierarchical refinement."""
    w, h, d = 1242, 375, 256
    x = synth.model_input_i8(w, h, d, 22)
    with api.StereoNetHIP(model_factory(w, h, d, multi=True), precision=prec) as eng:
        disp, raw = eng.infer(x)
    odisp, oraw, _ = oracle.forward(weights_multi, x, d)
    epe = float(np.abs(disp - odisp).mean())
    print(f"C5 multi 1242x375 D=256 prec={prec}: EPE {epe:.3e} px, max {np.abs(disp - odisp).max():.3e}")
    assert disp.shape == (h, w) and epe < EPE_TOL
    inv_q = np.float32(1.0 / (192.0 * float(np.float32(spec.OUT_SCALE))))
    assert (raw == np.rint(disp * inv_q).astype(np.int32)).all()


def test_multi_batch_chunks_pieces_and_async_equal_single(model_factory):
    """Ragged tower chunks / low-resolution pieces and the async slots run the same arithmetic per pair."""
    w, h, d = 200, 120, 64
    xs = np.stack([synth.model_input_i8(w, h, d, 40 + i) for i in range(7)])
    with api.StereoNetHIP(model_factory(w, h, d, multi=True), max_batch=7, precision=api.PREC_F16, refine_chunk=2,
                          piece=3) as eng:
        disp, raw = eng.infer(xs)
        singles = [eng.infer(xs[i]) for i in range(7)]
        araw = [np.empty((h, w), np.int32) for _ in range(4)]
        adisp = [np.empty((h, w), np.float32) for _ in range(4)]
        for rep in range(3):             # the third request per slot replays a captured hipGraph
            tickets = [eng.submit(xs[i], araw[i], adisp[i]) for i in range(4)]
            for t in tickets:
                eng.wait(t)
            for i in range(4):
                assert np.array_equal(araw[i], singles[i][1]) and np.array_equal(adisp[i], singles[i][0]), (rep, i)
    for i in range(7):
        assert np.array_equal(disp[i], singles[i][0]) and np.array_equal(raw[i], singles[i][1]), i


def test_multi_full_size_1280x720(model_factory, oracle, weights_multi):
    w, h, d = 1280, 720, 192
    x = synth.model_input_i8(w, h, d, 9)
    with api.StereoNetHIP(model_factory(w, h, d, multi=True), precision=api.PREC_F16) as eng:
        disp, raw = eng.infer(x)
        assert abs(eng.info.flops_per_pair / 1e9 - 295.56) < 0.01
    odisp, _, _ = oracle.forward(weights_multi, x, d)
    epe = float(np.abs(disp - odisp).mean())
    print(f"multi 1280x720 D=192 f16: EPE {epe:.3e} px")
    assert epe < EPE_TOL


def _random_cases():
    rng = np.random.default_rng(2026)
    cases = []
    for i in range(10):
        w = int(rng.integers(34, 330))
        h = int(rng.integers(18, 200))
        d = int(rng.choice([16, 32, 48, 64, 96, 128, 192, 256]))
        d = min(d, max(16, (w // 2) // 16 * 16))              # keep some columns with a full disparity range
        prec = [api.PREC_F16, api.PREC_FP32, api.PREC_F16X3][i % 3]
        cases.append((w, h, d, prec, bool(i % 2), int(rng.integers(1, 6))))
    return cases


@pytest.mark.parametrize("w,h,d,prec,multi,n", _random_cases())
def test_random_shapes_vs_oracle(model_factory, oracle, weights_blob, weights_multi, w, h, d, prec, multi, n):
    """Seeded random geometries (odd sizes, widths that are not multiples of 4 / 16 / 32 / 64, every D), batch sizes
    and both refinement forms against the oracle: exercises the tile-overhang paths of every kernel variant."""
    blob = weights_multi if multi else weights_blob
    xs = np.stack([synth.model_input_i8(w, h, d, 700 + i) for i in range(n)])
    with api.StereoNetHIP(model_factory(w, h, d, multi=multi), max_batch=n, precision=prec, refine_chunk=2, piece=3) as eng:
        disp, raw = eng.infer(xs)
        one, one_raw = eng.infer(xs[n - 1])
    disp = disp.reshape(n, h, w)
    raw = raw.reshape(n, h, w)
    assert np.array_equal(disp[n - 1], one.reshape(h, w)) and np.array_equal(raw[n - 1], one_raw.reshape(h, w))
    for i in (0, n - 1):
        odisp, oraw, _ = oracle.forward(blob, xs[i], d)
        err = np.abs(disp[i] - odisp)
        print(f"{w}x{h} D={d} prec={prec} multi={multi} n={n} pair {i}: EPE {err.mean():.2e} max {err.max():.2e}")
        assert err.mean() < EPE_TOL and np.isfinite(disp[i]).all()
        assert raw[i].min() >= 0


def test_large_geometry_and_refusal(model_factory, oracle, weights_blob):
    """A frame larger than any BASELINE config (2048x1088, D=256: 2.4x the pixels of 1280x720) against the oracle, and the
    size sn_create must refuse (a tensor would pass the 32-bit byte offsets the kernels use)."""
    w, h, d = 2048, 1088, 256
    x = synth.model_input_i8(w, h, d, 77)
    with api.StereoNetHIP(model_factory(w, h, d), precision=api.PREC_F16) as eng:
        assert eng.refine_chunk >= 1
        disp, raw = eng.infer(x)
    odisp, _, _ = oracle.forward(weights_blob, x, d)
    epe = float(np.abs(disp - odisp).mean())
    print(f"2048x1088 D=256 f16: EPE {epe:.3e} px")
    assert epe < EPE_TOL and raw.min() >= 0
    with pytest.raises(api.StereoNetError):
        api.StereoNetHIP(model_factory(w, h, d), width=16384, height=16384, precision=api.PREC_F16, refine_chunk=8,
                         max_batch=8)


_ENV_SCRIPT = r"""
import sys, numpy as np
sys.path.insert(0, sys.argv[1])
from hobot_stereonet_amd import api, synth
w, h, d, n = 200, 120, 64, 5
xs = np.stack([synth.model_input_i8(w, h, d, 40 + i) for i in range(n)])
with api.StereoNetHIP(sys.argv[2], max_batch=n, precision=api.PREC_F16, refine_chunk=2, piece=3) as eng:
    disp, raw = eng.infer(xs)
np.save(sys.argv[3], disp)
"""


@pytest.mark.parametrize("env,exact", [({"SN_TOWER_STREAMS": "2"}, True), ({"SN_NO_OVERLAP": "1"}, True), ({"SN_REV": "0"}, True),
                                       # the streamed blocks are bit-identical to the two-launch form and the tail form to block +
                                       # k_head_final_f16, so every way of mixing them changes nothing
                                       ({"SN_FUSE": "0"}, True), ({"SN_STREAM_DIL": "2"}, True), ({"SN_TAIL_FUSE": "0"}, True),
                                       ({"SN_TOWER_STREAMS": "2", "SN_TAIL_FUSE": "0"}, True), ({"SN_STREAM_WGS": "100"}, True),
                                       ({"SN_STREAM_PRIORITY": "1"}, True),
                                       # aggregation layers / down-convs on the plain tensors (k_conv_x3s) instead of the zero-bordered ones
                                       # (the plain-volume form has no folded output conv: SN_HEAD_FOLD=0 on both sides of those two)
                                       ({"SN_AGG_DMA": "0", "SN_HEAD_FOLD": "0"}, True), ({"SN_DOWN_DMA": "0"}, True),
                                       ({"SN_AGG_DMA": "0", "SN_DOWN_DMA": "0", "SN_HEAD_FOLD": "0"}, True),
                                       ({"SN_FEAT_DMA": "0"}, True),
                                       # the first two down-convs as two kernels instead of the folded 13x13 stride-4 conv (another summation
                                       # order of the same linear map: not bit-identical)
                                       ({"SN_DOWN01": "0"}, False), ({"SN_DOWN01": "0", "SN_DOWN_DMA": "0"}, False),
                                       # the aggregation network's output conv as its own kernel on the 32-channel volume (k_head_softargmin)
                                       # instead of the taps-as-M contraction in the last layer's epilogue + k_softargmin_p
                                       ({"SN_HEAD_FOLD": "0"}, False), ({"SN_HEAD_FOLD": "0", "SN_AGG_DMA": "0"}, False)])
def test_diagnostic_switches_run_the_same_network(model_factory, oracle, weights_blob, tmp_path, env, exact):
    """The library's diagnostic environment switches select other schedules / kernel pairings of the
    SAME arithmetic: stream layout switches must be bit-identical to the default, kernel pairings within the EPE bar."""
    import os
    import subprocess
    import sys
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    w, h, d, n = 200, 120, 64, 5
    model = model_factory(w, h, d)
    script = tmp_path / "run.py"
    script.write_text(_ENV_SCRIPT)
    outs = {}
    base = {"SN_HEAD_FOLD": "0"} if (exact and env.get("SN_HEAD_FOLD") == "0") else {}      # equality needs the same head form
    for tag, e in (("default", base), ("switch", env)):
        out = str(tmp_path / f"{tag}.npy")
        r = subprocess.run([sys.executable, str(script), root, model, out], env=dict(os.environ, **e), capture_output=True,
                           text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        outs[tag] = np.load(out).reshape(n, h, w)
    if exact:
        assert np.array_equal(outs["default"], outs["switch"])
    for i in (0, n - 1):
        odisp, _, _ = oracle.forward(weights_blob, synth.model_input_i8(w, h, d, 40 + i), d)
        assert np.abs(outs["switch"][i] - odisp).mean() < EPE_TOL
