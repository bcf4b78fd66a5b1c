"""HIP path vs the CPU oracle, through the C ABI (include/stereonet_hip.h).  Needs an MI355X.

Tolerances: byte/integer work (pre-processing, wire format) is bit-exact; the network is fp32
arithmetic in a different summation order, bounded by EPE <= 1e-3 px (BASELINE.json north_star)."""
import numpy as np
import pytest

from hobot_stereonet_amd import api, spec, synth

pytestmark = pytest.mark.gpu
EPE_TOL = 1e-3


@pytest.fixture(scope="module")
def small_engine(model_factory):
    eng = api.StereoNetHIP(model_factory(96, 64, 48), max_batch=2, precision=api.PREC_FP32)
    yield eng
    eng.close()


def rel_err(a, b):
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-12))


# ---- per-kernel parity ------------------------------------------------------------------------
@pytest.mark.parametrize("h,w,dil", [(45, 80, 1), (64, 96, 1), (33, 70, 2), (72, 200, 4), (130, 300, 8),
                                     (64, 128, 1), (20, 20, 8), (128, 257, 1), (190, 130, 2)])
def test_conv3x3_c32(small_engine, oracle, h, w, dil):
    rng = np.random.default_rng(h * 1000 + w + dil)
    x = rng.standard_normal((32, h, w)).astype(np.float32)
    wt = (rng.standard_normal((32, 32, 3, 3)) / 17.0).astype(np.float32)
    b = rng.standard_normal(32).astype(np.float32)
    ref = oracle.conv2d(x, wt, b, 1, dil, dil)
    got = small_engine.dbg_conv2d(x, wt, b, 3, 1, dil)
    assert rel_err(got, ref) < 2e-5
    # residual + LeakyReLU epilogue (the in-place res-block form)
    res = rng.standard_normal((32, h, w)).astype(np.float32)
    v = ref + res
    ref2 = np.where(v > 0, v, v * np.float32(0.2))
    got2 = small_engine.dbg_conv2d(x, wt, b, 3, 1, dil, lrelu=True, residual=res)
    assert rel_err(got2, ref2) < 2e-5


@pytest.mark.parametrize("h,w,dil", [(64, 128, 1), (64, 112, 1), (48, 176, 1), (45, 80, 1), (33, 72, 2), (72, 200, 4),
                                     (130, 304, 8), (20, 20, 8), (90, 160, 2), (720, 1280, 1)])
def test_tower_conv_fp32_kernel(small_engine, oracle, h, w, dil):
    """The weights-stationary fp32 tower kernel (k_ref_conv_f32, what SN_PREC_FP32 runs for the twelve 32 -> 32 tower
    layers) on full, partial (w % 64 != 0, h % 8 != 0) and single-tile geometries, plain and as the in-place
    residual + LeakyReLU form."""
    rng = np.random.default_rng(h * 1000 + w + dil)
    x = rng.standard_normal((32, h, w)).astype(np.float32)
    wt = (rng.standard_normal((32, 32, 3, 3)) / 17.0).astype(np.float32)
    b = rng.standard_normal(32).astype(np.float32)
    ref = oracle.conv2d(x, wt, b, 1, dil, dil)
    got = small_engine.dbg_conv2d(x, wt, b, 3, 1, dil, tower32=True)
    assert rel_err(got, ref) < 2e-5
    res = rng.standard_normal((32, h, w)).astype(np.float32)
    v = ref + res
    ref2 = np.where(v > 0, v, v * np.float32(0.2))
    got2 = small_engine.dbg_conv2d(x, wt, b, 3, 1, dil, lrelu=True, residual=res, tower32=True)
    assert rel_err(got2, ref2) < 2e-5


@pytest.mark.parametrize("cin,h,w", [(3, 64, 96), (3, 360, 640), (32, 90, 160), (32, 46, 82), (32, 180, 320)])
def test_conv5x5_stride2(small_engine, oracle, cin, h, w):
    rng = np.random.default_rng(cin * 7 + h + w)
    x = rng.standard_normal((cin, h, w)).astype(np.float32)
    wt = (rng.standard_normal((32, cin, 5, 5)) / np.sqrt(cin * 25)).astype(np.float32)
    b = rng.standard_normal(32).astype(np.float32)
    ref = oracle.conv2d(x, wt, b, 2, 2, 1)
    got = small_engine.dbg_conv2d(x, wt, b, 5, 2, 1)
    assert got.shape == ref.shape
    assert rel_err(got, ref) < 2e-5


def test_conv3x3_few_channels(small_engine, oracle):
    rng = np.random.default_rng(5)
    x = rng.standard_normal((4, 70, 150)).astype(np.float32)
    wt = rng.standard_normal((32, 4, 3, 3)).astype(np.float32)
    b = rng.standard_normal(32).astype(np.float32)
    assert rel_err(small_engine.dbg_conv2d(x, wt, b, 3, 1, 1), oracle.conv2d(x, wt, b, 1, 1, 1)) < 2e-5


@pytest.mark.parametrize("d,h,w", [(3, 4, 6), (12, 45, 80), (6, 17, 33)])
def test_conv3d(small_engine, oracle, d, h, w):
    rng = np.random.default_rng(d + h + w)
    x = rng.standard_normal((32, d, h, w)).astype(np.float32)
    wt = (rng.standard_normal((32, 32, 3, 3, 3)) / 30.0).astype(np.float32)
    b = rng.standard_normal(32).astype(np.float32)
    ref = oracle.conv3d(x, wt, b)
    got = small_engine.dbg_conv3d(x, wt, b)
    assert rel_err(got, ref) < 2e-5


# ---- pre-processing: bit-exact ---------------------------------------------------------------------
def test_preprocess_nv12_bit_exact(small_engine, oracle, golden_pre):
    for c in ("rand32x16", "rand64x36", "rand48x20", "ramp8x4"):
        w, h = map(int, golden_pre[c + ".wh"])
        left = golden_pre[c + ".nv12"]
        right = synth.random_nv12(w, h, 17)
        got = small_engine.preprocess_nv12(left, right, w, h)
        exp_l = (golden_pre[c + ".yuv444"].reshape(3, h, w) ^ np.uint8(0x80)).view(np.int8)
        assert (got[:3] == exp_l).all(), c                      # vs the reference's own output
        assert (got == oracle.preprocess_nv12(left, right, w, h)).all(), c


# ---- end to end --------------------------------------------------------------------------------------
CASES = [("c96x64_d48", 96, 64, 48, 3), ("c160x96_d96", 160, 96, 96, 4), ("c100x52_d32", 100, 52, 32, 5)]


@pytest.mark.parametrize("name,w,h,d,seed", CASES)
def test_forward_small_vs_golden_and_oracle(model_factory, oracle, golden_net, weights_blob, name, w, h, d, seed):
    x = synth.model_input_i8(w, h, d, seed)
    with api.StereoNetHIP(model_factory(w, h, d), precision=api.PREC_FP32) as eng:
        disp, raw = eng.infer(x)
        low = eng.dbg_read("disp_low").reshape((h + 15) // 16, (w + 15) // 16)
        cost = eng.dbg_read("cost").reshape(d // 16, (h + 15) // 16, (w + 15) // 16)
    odisp, oraw, olow = oracle.forward(weights_blob, x, d)
    assert np.abs(cost - golden_net[name + ".cost"]).max() < 2e-4 * max(1.0, np.abs(golden_net[name + ".cost"]).max())
    assert np.abs(low - olow).max() < 1e-4
    assert np.abs(disp - golden_net[name + ".disp"]).mean() < EPE_TOL
    assert np.abs(disp - odisp).mean() < EPE_TOL
    assert np.abs(disp - odisp).max() < 20 * EPE_TOL
    # wire format is the same integer map of the float disparity as the oracle's
    inv_q = np.float32(1.0 / (192.0 * float(np.float32(spec.OUT_SCALE))))
    assert (raw == np.rint(disp * inv_q).astype(np.int32)).all()
    assert raw.min() >= 0


def test_features_identical_eyes(model_factory, oracle, weights_blob):
    # reference fixture config/image_left.jpg == image_right.jpg: equal eyes -> equal feature maps
    w, h, d = 96, 64, 48
    x = synth.model_input_i8(w, h, d, 5).copy()
    x[3:] = x[:3]
    with api.StereoNetHIP(model_factory(w, h, d), precision=api.PREC_FP32) as eng:
        eng.infer(x)
        fl, fr = eng.dbg_read("feat_l"), eng.dbg_read("feat_r")
    assert (fl == fr).all()
    planes = x[:3].astype(np.float32) / 128.0
    assert rel_err(fl.reshape(32, 4, 6), oracle.features(weights_blob, planes)) < 5e-5


@pytest.mark.parametrize("prec", [api.PREC_FP32, api.PREC_F16])
def test_identical_eyes_reference_fixture(model_factory, oracle, weights_blob, prec):
    """The reference's only image fixture: config/image_left.jpg == image_right.jpg (byte-identical files).  A 160x96
    crop of it (tests/golden/identical_eyes.npz, made by make_identical_eyes_golden.py) goes through the offline feeder's
    steps (BGR -> NV12 -> CvtNV12Data2Tensors) for both eyes.  Known answers: the feature maps of the eyes are equal bit
    for bit, so the cost volume is EXACTLY zero wherever x >= d; and the HIP path still matches the oracle."""
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "identical_eyes.npz"))
    assert str(g["source_sha256"]) == "fde9f79716b34223705137cad4c4e027df3bc6f765e1793d65fe6895a33c6b08"
    bgr = g["bgr"]
    h, w = bgr.shape[:2]
    d = 96
    nv12 = oracle.bgr_to_nv12(bgr)
    x = oracle.preprocess_nv12(nv12, nv12, w, h)
    assert (x[:3] == x[3:]).all()
    with api.StereoNetHIP(model_factory(w, h, d), precision=prec) as eng:
        disp, raw = eng.infer(x)
        fl, fr = eng.dbg_read("feat_l"), eng.dbg_read("feat_r")
    assert (fl == fr).all()                                      # Siamese tower: identical inputs, identical outputs
    fl = fl.reshape(32, h // 16, w // 16)
    cv = oracle.cost_volume(fl, fl, d // 16)                     # fL - shift(fL): plane 0 is exactly zero
    assert (cv[:, 0] == 0).all()
    odisp, oraw, _ = oracle.forward(weights_blob, x, d)
    assert np.abs(disp - odisp).mean() < EPE_TOL
    assert raw.min() >= 0


def test_full_size_epe(model_factory, oracle, weights_blob):
    """BASELINE.json configs[1]: 1280x720, D=192, one pair, fp32."""
    w, h, d = 1280, 720, 192
    x = synth.model_input_i8(w, h, d, 0)
    with api.StereoNetHIP(model_factory(w, h, d), precision=api.PREC_FP32) as eng:
        disp, raw = eng.infer(x)
        disp2, raw2 = eng.infer(x)
    odisp, oraw, _ = oracle.forward(weights_blob, x, d)
    epe = float(np.abs(disp - odisp).mean())
    print(f"EPE vs oracle at 1280x720 D=192: {epe:.3e} px, max {np.abs(disp - odisp).max():.3e}")
    assert epe < EPE_TOL
    assert (disp == disp2).all() and (raw == raw2).all()          # deterministic
    assert np.abs(raw.astype(np.int64) - oraw).max() <= 1 + int(20 * EPE_TOL / (192 * spec.OUT_SCALE))
    # the render node's dequantisation (publisher_member_function.py:65-75) recovers the disparity
    back = raw.view(np.uint32).astype(np.float64) * spec.OUT_SCALE * 16 * 12
    assert np.abs(back - disp).max() < 0.51 * 192 * spec.OUT_SCALE + 1e-5


def test_padded_geometry(model_factory, oracle, weights_blob):
    # sizes that are not multiples of 16 are zero-padded right/bottom and cropped (C1 960x540, C5 1242x375 shapes)
    w, h, d = 124, 38, 32
    x = synth.model_input_i8(w, h, d, 9)
    with api.StereoNetHIP(model_factory(w, h, d), precision=api.PREC_FP32) as eng:
        disp, _ = eng.infer(x)
    odisp, _, _ = oracle.forward(weights_blob, x, d)
    assert disp.shape == (h, w)
    assert np.abs(disp - odisp).mean() < EPE_TOL


def test_batch_equals_single_and_async(small_engine):
    w, h, d = 96, 64, 48
    xs = np.stack([synth.model_input_i8(w, h, d, s) for s in (1, 2)])
    disp_b, raw_b = small_engine.infer(xs)
    for i in range(2):
        dsp, rw = small_engine.infer(xs[i])
        assert (dsp == disp_b[i]).all() and (rw == raw_b[i]).all()     # bit-identical: no cross-pair math
    # async Run (is_sync_mode=false, stereonet_node.cpp:812): task_num=4 in flight
    outs = [(np.empty((h, w), np.int32), np.empty((h, w), np.float32)) for _ in range(6)]
    tickets = []
    for i in range(6):
        if len(tickets) == 4:
            small_engine.wait(tickets.pop(0))
        tickets.append(small_engine.submit(xs[i % 2], outs[i][0], outs[i][1]))
    for t in tickets:
        assert small_engine.wait(t) > 0.0
    for i in range(6):
        assert (outs[i][0] == raw_b[i % 2]).all() and (outs[i][1] == disp_b[i % 2]).all()
    with pytest.raises(api.StereoNetError):
        small_engine.wait(12345)


def test_side_by_side_nv12_path(model_factory, oracle):
    # FeedImg split + CvtNV12Data2Tensors + Run fused on the device == the reference's three host steps
    w, h, d = 96, 64, 48
    sbs = np.random.default_rng(4).integers(0, 256, (h * 3 // 2) * 2 * w, dtype=np.uint8)
    left, right = oracle.split_sbs_nv12(sbs, w, h)
    ten_ref = oracle.preprocess_nv12(left, right, w, h)
    with api.StereoNetHIP(model_factory(w, h, d), precision=api.PREC_FP32) as eng:
        disp, raw, ten = eng.infer_sbs_nv12(sbs, want_tensor=True)
        disp2, raw2 = eng.infer(ten_ref)
    assert (ten == ten_ref).all()
    assert (disp == disp2).all() and (raw == raw2).all()


def test_async_and_batched_nv12_ingest(model_factory, oracle):
    """sn_submit_nv12 (async Run on FeedImg's raw frame, graph-replayed from the third use of a slot on) and
    sn_preprocess_sbs_nv12_batch (the streaming ingest) == the reference's host split + CvtNV12Data2Tensors, then Run."""
    w, h, d = 96, 64, 48
    rng = np.random.default_rng(14)
    frames = [rng.integers(0, 256, (h * 3 // 2) * 2 * w, dtype=np.uint8) for _ in range(3)]
    tens = np.stack([oracle.preprocess_nv12(*oracle.split_sbs_nv12(f, w, h), w, h) for f in frames])
    with api.StereoNetHIP(model_factory(w, h, d), precision=api.PREC_F16, max_batch=3, task_num=2) as eng:
        disp_ref, raw_ref = eng.infer(tens)
        got = eng.preprocess_sbs_nv12(np.stack(frames))
        assert (got == tens).all()
        outs = [(np.empty((h, w), np.int32), np.empty((h, w), np.float32)) for _ in range(9)]
        tickets = []
        for i in range(9):                  # 2 slots x (plain, capture, replay, replay...) and a mix with sn_submit
            if len(tickets) == 2:
                eng.wait(tickets.pop(0))
            if i == 4:
                tickets.append(eng.submit(tens[i % 3], outs[i][0], outs[i][1]))
            else:
                tickets.append(eng.submit_nv12(frames[i % 3], outs[i][0], outs[i][1]))
        for t in tickets:
            eng.wait(t)
        for i in range(9):
            assert (outs[i][0] == raw_ref[i % 3]).all() and (outs[i][1] == disp_ref[i % 3]).all(), i
        with pytest.raises(api.StereoNetError):
            eng.submit_nv12(frames[0][:-2], outs[0][0], None)


def test_error_behaviour(model_factory, tmp_path):
    with pytest.raises(api.StereoNetError) as e:
        api.StereoNetHIP(str(tmp_path / "missing.snw"))
    assert e.value.code == -2          # reference: access() check fails -> Init returns -1 (stereonet_node.cpp:131-134)
    bad = tmp_path / "bad.snw"
    bad.write_bytes(b"not a model" * 10)
    with pytest.raises(api.StereoNetError) as e:
        api.StereoNetHIP(str(bad))
    assert e.value.code == -3
    with api.StereoNetHIP(model_factory(96, 64, 48), precision=api.PREC_FP32) as eng:
        with pytest.raises(api.StereoNetError):
            eng.infer(np.zeros((6, 32, 32), np.int8))           # geometry mismatch (stereonet_node.cpp:682-690)
        with pytest.raises(api.StereoNetError):
            eng.infer(np.zeros((3, 6, 64, 96), np.int8))        # n > max_batch


@pytest.mark.parametrize("prec", [api.PREC_FP32, api.PREC_F16])
def test_baseline_config_c1_960x540_d48(model_factory, oracle, weights_blob, prec):
    """BASELINE.json configs[0] shape: one 960x540 pair, D=48 (H is padded to 544 internally and cropped)."""
    w, h, d = 960, 540, 48
    x = synth.model_input_i8(w, h, d, 21)
    with api.StereoNetHIP(model_factory(w, h, d), precision=prec) as eng:
        disp, raw = eng.infer(x)
    odisp, oraw, _ = oracle.forward(weights_blob, x, d)
    epe = float(np.abs(disp - odisp).mean())
    print(f"C1 960x540 D=48 prec={prec}: EPE {epe:.3e} px")
    assert disp.shape == (h, w) and epe < EPE_TOL


@pytest.mark.parametrize("prec", [api.PREC_FP32, api.PREC_F16])
def test_baseline_config_c5_kitti_1242x375_d256(model_factory, oracle, weights_blob, prec):
    """BASELINE.json configs[4] shape: KITTI-2015 1242x375, D=256 (16 planes; padded to 1248x384)."""
    w, h, d = 1242, 375, 256
    x = synth.model_input_i8(w, h, d, 22)
    with api.StereoNetHIP(model_factory(w, h, d), precision=prec) as eng:
        disp, raw = eng.infer(x)
    odisp, oraw, _ = oracle.forward(weights_blob, x, d)
    epe = float(np.abs(disp - odisp).mean())
    print(f"C5 1242x375 D=256 prec={prec}: EPE {epe:.3e} px")
    assert disp.shape == (h, w) and epe < EPE_TOL
    inv_q = np.float32(1.0 / (192.0 * float(np.float32(spec.OUT_SCALE))))
    assert (raw == np.rint(disp * inv_q).astype(np.int32)).all()


@pytest.mark.parametrize("h,w", [(45, 80), (34, 60), (24, 78), (64, 96), (9, 33), (68, 128), (3, 5)])
def test_lowres_split_conv3x3(small_engine, oracle, h, w):
    """Low-resolution 3x3 layers of the fp16 modes: 22-bit split operands (k_conv_x3s) on hi/lo fp16 slot tensors (the
    production format; the hook converts), so the input is rounded to 22 bits and the output carries one more 22-bit
    rounding."""
    rng = np.random.default_rng(h * 5 + w)
    x = rng.standard_normal((32, h, w)).astype(np.float32)
    wt = (rng.standard_normal((32, 32, 3, 3)) / 17.0).astype(np.float32)
    b = rng.standard_normal(32).astype(np.float32)
    ref = oracle.conv2d(x, wt, b, 1, 1, 1)
    got = small_engine.dbg_conv2d(x, wt, b, 3, 1, 1, x3=True, slots=True)
    assert rel_err(got, ref) < 4e-6
    res = rng.standard_normal((32, h, w)).astype(np.float32)
    v = ref + res
    ref2 = np.where(v > 0, v, v * np.float32(0.2))
    got2 = small_engine.dbg_conv2d(x, wt, b, 3, 1, 1, lrelu=True, residual=res, x3=True, slots=True)
    assert rel_err(got2, ref2) < 4e-6
    # the same layer on zero-bordered tensors (k_feat_x3s_dma: LDS-DMA staging, every wave holds both channel chunks, one
    # barrier per tile): the two chunks' sums are formed separately and added as k_conv_x3s's two K halves are, so the
    # same bits — plain, and with the in-place residual + activation
    got_dma = small_engine.dbg_conv2d(x, wt, b, 3, 1, 1, x3=True, slots=True, dma=True)
    assert np.array_equal(got_dma, got)
    got2_dma = small_engine.dbg_conv2d(x, wt, b, 3, 1, 1, lrelu=True, residual=res, x3=True, slots=True, dma=True)
    assert np.array_equal(got2_dma, got2)


@pytest.mark.parametrize("d,h,w", [(3, 4, 6), (12, 45, 80), (16, 24, 78), (1, 8, 16), (6, 23, 40), (12, 90, 160)])
def test_lowres_split_conv3d(small_engine, oracle, d, h, w):
    rng = np.random.default_rng(d + h + w)
    x = rng.standard_normal((32, d, h, w)).astype(np.float32)
    wt = (rng.standard_normal((32, 32, 3, 3, 3)) / 30.0).astype(np.float32)
    b = rng.standard_normal(32).astype(np.float32)
    ref = oracle.conv3d(x, wt, b)
    got = small_engine.dbg_conv3d(x, wt, b, x3=True, slots=True)
    assert rel_err(got, ref) < 4e-6
    # the same layer on zero-bordered volumes (LDS-DMA staging, double-buffered tile, deferred epilogue): the same MFMAs
    # in the same order, so the same bits — with and without the activation
    got_dma = small_engine.dbg_conv3d(x, wt, b, x3=True, slots=True, dma=True)
    assert np.array_equal(got_dma, got)
    a = small_engine.dbg_conv3d(x, wt, b, lrelu=True, x3=True, slots=True)
    a_dma = small_engine.dbg_conv3d(x, wt, b, lrelu=True, x3=True, slots=True, dma=True)
    assert np.array_equal(a_dma, a)


@pytest.mark.parametrize("h,w", [(90, 160), (46, 82), (360, 640), (64, 96), (8, 32), (2, 2), (30, 70)])
def test_lowres_split_conv5x5_stride2(small_engine, oracle, h, w):
    rng = np.random.default_rng(h * 3 + w)
    x = rng.standard_normal((32, h, w)).astype(np.float32)
    wt = (rng.standard_normal((32, 32, 5, 5)) / 28.0).astype(np.float32)
    b = rng.standard_normal(32).astype(np.float32)
    ref = oracle.conv2d(x, wt, b, 2, 2, 1)
    got = small_engine.dbg_conv2d(x, wt, b, 5, 2, 1, x3=True, slots=True)
    assert got.shape == ref.shape
    assert rel_err(got, ref) < 4e-6
    # the same layer on zero-bordered tensors (LDS-DMA staging, 4 x 16 double-buffered tiles, deferred epilogue): the same
    # K order and MFMAs per output, so the same bits
    got_dma = small_engine.dbg_conv2d(x, wt, b, 5, 2, 1, x3=True, slots=True, dma=True)
    assert np.array_equal(got_dma, got)
    a = small_engine.dbg_conv2d(x, wt, b, 5, 2, 1, lrelu=True, x3=True, slots=True)
    a_dma = small_engine.dbg_conv2d(x, wt, b, 5, 2, 1, lrelu=True, x3=True, slots=True, dma=True)
    assert np.array_equal(a_dma, a)


@pytest.mark.gpu
def test_device_depth_matches_host_parse_bit_for_bit(model_factory):
    """sn_depth_from_raw = Parse()'s dequantisation + depth (parser.cpp:84-86) as a kernel: the reference's float / double
    mix reproduced, so the GPU map equals the host twin's arithmetic bit for bit, including raw = 0 -> inf and the KAT of
    SURVEY §8(c) (int32 200000 -> 0.632 m)."""
    w, h, d = 96, 64, 48
    rng = np.random.default_rng(12)
    raw = rng.integers(0, 400000, (2, h, w)).astype(np.int32)
    raw[0, 0, :4] = [0, 1, 200000, 2 ** 31 - 1]
    with api.StereoNetHIP(model_factory(w, h, d), max_batch=2) as eng:
        depth, disp = eng.depth_from_raw(raw, want_disp=True)
        scale = np.float32(eng.out_scale)
    f, B = np.float32(527.1931762695312), np.float32(119.89382172)
    dis = raw.astype(np.float32) * scale                                           # float
    with np.errstate(divide="ignore"):
        ref = (np.float64(f * B) / (dis.astype(np.float64) * 16.0 * 12.0) / 1000.0).astype(np.float32)
    assert np.array_equal(depth, ref) and np.isinf(depth[0, 0, 0])
    assert abs(float(depth[0, 0, 2]) - 0.632) < 1e-3
    assert np.array_equal(disp, dis * np.float32(16.0) * np.float32(12.0))
