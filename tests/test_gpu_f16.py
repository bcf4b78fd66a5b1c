"""fp16 refinement tower (SN_PREC_F16) vs the CPU oracle.  Operands are rounded to fp16, products are
exact and accumulation is fp32, so a single layer must match the oracle run on fp16-rounded operands to
fp32 round-off + one output rounding; end to end the bound is the north-star's EPE <= 1e-3 px."""
import numpy as np
import pytest

from hobot_stereonet_amd import api, spec, synth

pytestmark = pytest.mark.gpu
EPE_TOL = 1e-3


def q16(a):
    return np.asarray(a, np.float32).astype(np.float16).astype(np.float32)


@pytest.fixture(scope="module")
def eng16(model_factory):
    eng = api.StereoNetHIP(model_factory(96, 64, 48), max_batch=2, precision=api.PREC_F16)
    yield eng
    eng.close()


@pytest.mark.parametrize("h,w,dil,tw", [(16, 64, 1, 64), (64, 96, 1, 32), (45, 80, 2, 64), (45, 80, 2, 32), (72, 200, 4, 0),
                                        (130, 300, 8, 0), (8, 64, 8, 0), (100, 129, 1, 64), (100, 129, 1, 32), (24, 70, 4, 0),
                                        (720, 1280, 1, 64), (720, 1280, 1, 32), (720, 1280, 2, 0)])
def test_ref_conv_f16_layer(eng16, oracle, h, w, dil, tw):
    """tw: tile width of the dilation-1 / -2 kernel (8x64 or 8x32; 0 = what the engine picks for this launch)"""
    rng = np.random.default_rng(h * 31 + w + dil)
    x = q16(rng.standard_normal((32, h, w)))
    wt = q16(rng.standard_normal((32, 32, 3, 3)) / 17.0)
    b = rng.standard_normal(32).astype(np.float32)
    ref = oracle.conv2d(x, wt, b, 1, dil, dil)
    got = eng16.dbg_ref_conv_f16(x, wt, b, dil, tile_w=tw)
    scale = np.abs(ref).max()
    assert np.abs(got - q16(ref)).max() <= 2e-3 * scale / 2 + 1e-6      # <= 1 fp16 ulp of the largest value
    assert np.abs(got - ref).mean() < 3e-4 * scale
    res = q16(rng.standard_normal((32, h, w)))
    v = ref + res
    ref2 = np.where(v > 0, v, v * np.float32(0.2))
    got2 = eng16.dbg_ref_conv_f16(x, wt, b, dil, lrelu=True, residual=res, tile_w=tw)
    assert np.abs(got2 - ref2).max() <= 1.2e-3 * np.abs(ref2).max() + 1e-6
    if tw:             # both tile shapes compute the same sums in the same order: bit-identical
        other = eng16.dbg_ref_conv_f16(x, wt, b, dil, lrelu=True, residual=res, tile_w=96 - tw)
        assert np.array_equal(got2, other)


CASES = [("c96x64_d48", 96, 64, 48, 3), ("c160x96_d96", 160, 96, 96, 4), ("c100x52_d32", 100, 52, 32, 5)]


@pytest.mark.parametrize("name,w,h,d,seed", CASES)
def test_forward_small_f16(model_factory, oracle, golden_net, weights_blob, name, w, h, d, seed):
    x = synth.model_input_i8(w, h, d, seed)
    with api.StereoNetHIP(model_factory(w, h, d), precision=api.PREC_F16) as eng:
        disp, raw = eng.infer(x)
        low = eng.dbg_read("disp_low").reshape((h + 15) // 16, (w + 15) // 16)
        cost = eng.dbg_read("cost").reshape(d // 16, (h + 15) // 16, (w + 15) // 16)
    odisp, _, olow = oracle.forward(weights_blob, x, d)
    assert np.abs(low - olow).max() < 1e-4            # the low-resolution branch: 22-bit split operands
    # the matching costs themselves: the output conv of the aggregation network rides on the last layer's epilogue as a
    # taps-as-M contraction (k_agg_x3s_dma HEADP) and k_softargmin_p sums its 27 shifted partial sums
    gcost = golden_net[name + ".cost"]
    assert np.abs(cost - gcost).max() < 2e-4 * max(1.0, np.abs(gcost).max())
    epe = float(np.abs(disp - odisp).mean())
    assert epe < EPE_TOL, epe
    assert np.abs(disp - golden_net[name + ".disp"]).mean() < EPE_TOL
    inv_q = np.float32(1.0 / (192.0 * float(np.float32(spec.OUT_SCALE))))
    assert (raw == np.rint(disp * inv_q).astype(np.int32)).all()


def test_full_size_epe_f16(model_factory, oracle, weights_blob):
    """BASELINE.json configs[2] shape: 1280x720 D=192 on the fp16 MFMA path, batch 2, refine_chunk 2."""
    w, h, d = 1280, 720, 192
    xs = np.stack([synth.model_input_i8(w, h, d, s) for s in (0, 1)])
    with api.StereoNetHIP(model_factory(w, h, d), precision=api.PREC_F16, max_batch=2, refine_chunk=2) as eng:
        disp, raw = eng.infer(xs)
        disp_again, _ = eng.infer(xs)
        d0, _ = eng.infer(xs[0])
    assert (disp == disp_again).all()                  # deterministic, border untouched between calls
    assert (d0 == disp[0]).all()                       # batch == single
    for i in range(2):
        odisp, _, _ = oracle.forward(weights_blob, xs[i], d)
        epe = float(np.abs(disp[i] - odisp).mean())
        print(f"fp16 tower EPE vs oracle, pair {i}: {epe:.3e} px (max {np.abs(disp[i] - odisp).max():.3e})")
        assert epe < EPE_TOL


def test_padded_geometry_f16(model_factory, oracle, weights_blob):
    w, h, d = 124, 38, 32
    x = synth.model_input_i8(w, h, d, 9)
    with api.StereoNetHIP(model_factory(w, h, d), precision=api.PREC_F16) as eng:
        disp, _ = eng.infer(x)
    odisp, _, _ = oracle.forward(weights_blob, x, d)
    assert np.abs(disp - odisp).mean() < EPE_TOL


@pytest.mark.parametrize("prec", [api.PREC_F16, api.PREC_FP32])
def test_two_stream_piece_pipeline_is_bit_identical(model_factory, prec):
    """n > piece switches on the two-stream pipeline (low-res of piece k+1 overlaps refinement of piece k);
    results must equal the one-pair-at-a-time results bit for bit, including a ragged last piece."""
    w, h, d = 160, 96, 96
    xs = np.stack([synth.model_input_i8(w, h, d, 20 + s) for s in range(7)])
    with api.StereoNetHIP(model_factory(w, h, d), precision=prec, max_batch=7, refine_chunk=2, piece=2) as eng:
        disp, raw = eng.infer(xs)
        disp2, raw2 = eng.infer(xs)
        singles = [eng.infer(xs[i]) for i in range(7)]
    assert (disp == disp2).all() and (raw == raw2).all()
    for i in range(7):
        assert (singles[i][0] == disp[i]).all() and (singles[i][1] == raw[i]).all(), i


@pytest.mark.parametrize("rc,piece,n", [(3, 4, 9), (3, 8, 11), (2, 3, 7), (1, 8, 9)])
def test_ragged_chunks_get_their_own_tile_queue(model_factory, rc, piece, n):
    """piece % refine_chunk != 0: every tower chunk of one forward() needs its own tile-queue counters (they are
    zeroed once per forward and never reset by the kernel).  With shared counters the second chunk sees an exhausted
    queue and leaves tiles uncomputed -> stale activations.  The image is large enough (> 2 rounds of tiles per XCD
    band) for the dynamic queue to be in use."""
    w, h, d = 1280, 720, 192
    xs = np.stack([synth.model_input_i8(w, h, d, 40 + (s % 3)) for s in range(n)])
    with api.StereoNetHIP(model_factory(w, h, d), precision=api.PREC_F16, max_batch=n, refine_chunk=rc, piece=piece) as eng:
        disp, raw = eng.infer(xs)
        singles = [eng.infer(xs[i]) for i in range(3)]
    for i in range(n):
        assert (singles[i % 3][0] == disp[i]).all() and (singles[i % 3][1] == raw[i]).all(), i


@pytest.mark.parametrize("fused", [0, 2])
@pytest.mark.parametrize("h,w,dil", [(8, 62, 1), (64, 96, 1), (45, 80, 1), (100, 129, 1), (37, 250, 1), (720, 1280, 1),
                                     (40, 70, 2), (45, 131, 2), (375, 1242, 2), (50, 140, 4), (375, 1242, 4), (37, 260, 8),
                                     (720, 1280, 8)])
def test_residual_block_f16(eng16, oracle, h, w, dil, fused):
    """y = lrelu(x + conv2(lrelu(conv1(x)+b1)) + b2).  fused = 0: two launches; 2: the row-streaming fused kernel (the
    pipeline's default).  Reference: oracle convs on fp16-rounded operands with t rounded to fp16."""
    rng = np.random.default_rng(h * 7 + w + dil)
    x = q16(rng.standard_normal((32, h, w)))
    w1 = q16(rng.standard_normal((32, 32, 3, 3)) / 17.0)
    w2 = q16(rng.standard_normal((32, 32, 3, 3)) / 17.0)
    b1 = rng.standard_normal(32).astype(np.float32)
    b2 = rng.standard_normal(32).astype(np.float32)
    t = oracle.conv2d(x, w1, b1, 1, dil, dil)
    t = q16(np.where(t > 0, t, t * np.float32(0.2)))
    v = x + oracle.conv2d(t, w2, b2, 1, dil, dil)
    ref = np.where(v > 0, v, v * np.float32(0.2))
    got = eng16.dbg_ref_block_f16(x, w1, b1, w2, b2, dil, fused)
    scale = np.abs(ref).max()
    # t may round differently by one fp16 ulp at a few positions (fp32 summation order), then one output rounding
    assert np.abs(got - ref).max() <= 3e-3 * scale
    assert np.abs(got - ref).mean() < 3e-4 * scale
    if fused == 2:      # same MFMA sequence per output as the two-launch form: bit-identical
        assert (got == eng16.dbg_ref_block_f16(x, w1, b1, w2, b2, dil, 0)).all()


def test_race_screen_full_size(model_factory):
    """Screen for rare races in the hand-synchronised kernels (LDS-DMA ring, counted vmcnt, raw barriers, two
    streams): the same 1280x720 batch must come out bit-identical on every one of 12 runs, and distinct inputs
    interleaved between the runs must not disturb it."""
    w, h, d = 1280, 720, 192
    xs = np.stack([synth.model_input_i8(w, h, d, 40 + s) for s in range(6)])
    other = np.stack([synth.model_input_i8(w, h, d, 60 + s) for s in range(6)])
    with api.StereoNetHIP(model_factory(w, h, d), precision=api.PREC_F16, max_batch=6, refine_chunk=2, piece=2) as eng:
        ref_disp, ref_raw = eng.infer(xs)
        for it in range(12):
            if it % 3 == 0:
                eng.infer(other)
            dsp, rw = eng.infer(xs)
            assert (rw == ref_raw).all() and (dsp == ref_disp).all(), it


# ---- SN_PREC_F16X3: hi/lo split operands, three fp16 MFMAs per product (fp32-class accuracy) -----------------
@pytest.mark.parametrize("h,w,dil", [(16, 64, 1), (45, 80, 2), (72, 200, 4), (130, 300, 8), (8, 64, 8), (100, 129, 1),
                                     (24, 70, 4), (720, 1280, 1)])
def test_ref_conv_f16x3_layer(eng16, oracle, h, w, dil):
    rng = np.random.default_rng(h * 13 + w + dil)
    x = rng.standard_normal((32, h, w)).astype(np.float32)
    wt = (rng.standard_normal((32, 32, 3, 3)) / 17.0).astype(np.float32)
    b = rng.standard_normal(32).astype(np.float32)
    ref = oracle.conv2d(x, wt, b, 1, dil, dil)
    got = eng16.dbg_ref_conv_f16x3(x, wt, b, dil)
    scale = np.abs(ref).max()
    # operands carry 22 bits, the result is stored as a 22-bit pair: a few 1e-6 relative to the largest value
    assert np.abs(got - ref).max() <= 6e-6 * scale
    res = rng.standard_normal((32, h, w)).astype(np.float32)
    v = ref + res
    ref2 = np.where(v > 0, v, v * np.float32(0.2))
    got2 = eng16.dbg_ref_conv_f16x3(x, wt, b, dil, lrelu=True, residual=res)
    assert np.abs(got2 - ref2).max() <= 6e-6 * np.abs(ref2).max()


@pytest.mark.parametrize("name,w,h,d,seed", CASES)
def test_forward_small_f16x3(model_factory, oracle, weights_blob, name, w, h, d, seed):
    x = synth.model_input_i8(w, h, d, seed)
    with api.StereoNetHIP(model_factory(w, h, d), precision=api.PREC_F16X3) as eng:
        disp, raw = eng.infer(x)
    odisp, _, _ = oracle.forward(weights_blob, x, d)
    epe = float(np.abs(disp - odisp).mean())
    assert epe < 1e-4, epe          # same class as the exact-fp32 path


def test_full_size_epe_f16x3(model_factory, oracle, weights_blob):
    w, h, d = 1280, 720, 192
    xs = np.stack([synth.model_input_i8(w, h, d, s) for s in (0, 1, 2)])
    with api.StereoNetHIP(model_factory(w, h, d), precision=api.PREC_F16X3, max_batch=3, refine_chunk=2, piece=2) as eng:
        disp, raw = eng.infer(xs)
        disp2, _ = eng.infer(xs)
        d0, _ = eng.infer(xs[0])
    assert (disp == disp2).all() and (d0 == disp[0]).all()
    odisp, _, _ = oracle.forward(weights_blob, xs[0], d)
    epe = float(np.abs(disp[0] - odisp).mean())
    print(f"f16x3 tower EPE vs oracle: {epe:.3e} px (max {np.abs(disp[0] - odisp).max():.3e})")
    assert epe < 2e-4


@pytest.mark.parametrize("h,w,tc", [(32, 64, 32), (64, 96, 32), (96, 160, 32), (52, 100, 32), (52, 102, 32),
                                    (90, 130, 32), (720, 1280, 32)])
def test_down0_f16_kernel(eng16, oracle, h, w, tc):
    """First down-conv (3->32, 5x5, stride 2) on the fp16 MFMA with densely packed K: the int8 input / 128 is
    exact in fp16 and the weights are split hi/lo (22 bits), so it must match the fp32 oracle to fp32 round-off
    of a 75-term sum; ragged sizes exercise the byte-load path (w % 4 != 0) and the padded output region."""
    rng = np.random.default_rng(h * 7 + w + tc)
    x = rng.integers(-128, 128, (6, h, w), dtype=np.int8)
    wt = (rng.standard_normal((32, 3, 5, 5)) / 8.0).astype(np.float32)
    b = rng.standard_normal(32).astype(np.float32)
    got = eng16.dbg_down0(x, wt, b, tc)
    hp, wp = (h + 15) // 16 * 16, (w + 15) // 16 * 16
    assert got.shape == (2, 32, hp // 2, wp // 2)
    for eye in range(2):
        xin = np.zeros((3, hp, wp), np.float32)               # the padded region is zero input (as in the pipeline)
        xin[:, :h, :w] = x[3 * eye:3 * eye + 3].astype(np.float32) / 128.0
        ref = oracle.conv2d(xin, wt, b, 2, 2, 1)
        assert ref.shape == got[eye].shape
        assert np.abs(got[eye] - ref).max() <= 2e-5 * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("h,w", [(16, 16), (32, 64), (64, 96), (96, 160), (52, 100), (52, 102), (90, 131), (375, 1242),
                                 (720, 1280)])
def test_down01_folded_kernels(eng16, oracle, h, w):
    """The first TWO down-convs as one 13x13 stride-4 convolution (csrc/sn_down01.hpp: k_down01_f16 for the inner weight
    class, k_down01_border for the first / last row and column): the int8 input / 128 is exact in fp16 and the folded
    weights are split hi/lo (22 bits), so the result must match the fp32 oracle of the two layers to fp32 round-off of an
    800-term sum of 75-term sums.  Ragged sizes exercise the unaligned / per-byte staging, partial tiles and the zero
    region right of / below the image."""
    rng = np.random.default_rng(h * 11 + w)
    x = rng.integers(-128, 128, (6, h, w), dtype=np.int8)
    w0 = (rng.standard_normal((32, 3, 5, 5)) / 8.0).astype(np.float32)
    b0 = rng.standard_normal(32).astype(np.float32)
    w1 = (rng.standard_normal((32, 32, 5, 5)) / 28.0).astype(np.float32)
    b1 = rng.standard_normal(32).astype(np.float32)
    got = eng16.dbg_down01(x, w0, b0, w1, b1)
    hp, wp = (h + 15) // 16 * 16, (w + 15) // 16 * 16
    assert got.shape == (2, 32, hp // 4, wp // 4)
    for eye in range(2):
        xin = np.zeros((3, hp, wp), np.float32)               # the padded region is zero input (as in the pipeline)
        xin[:, :h, :w] = x[3 * eye:3 * eye + 3].astype(np.float32) / 128.0
        ref = oracle.conv2d(oracle.conv2d(xin, w0, b0, 2, 2, 1), w1, b1, 2, 2, 1)
        assert ref.shape == got[eye].shape
        err = np.abs(got[eye] - ref)
        tol = 3e-5 * max(1.0, np.abs(ref).max())
        assert err[:, 1:-1, 1:-1].max() <= tol, "inner class"
        assert err.max() <= tol, "border classes"


@pytest.mark.parametrize("h,w,split", [(32, 64, False), (64, 96, True), (52, 100, False), (90, 130, True),
                                       (96, 160, False), (720, 1280, False), (720, 1280, True)])
def test_refin_f16_kernel(eng16, oracle, h, w, split):
    """Refinement input conv (4->32, 3x3, LeakyReLU) with the upsample / int8 planes produced in the loader:
    fp16 MFMA over 16-byte pixel slots [d_hi, Y, U, V, d_lo, 0, 0, 0] with split weights -> 22-bit operands.
    The fp16 (or hi/lo) output tensor is compared with the fp32 oracle of the same layer."""
    rng = np.random.default_rng(h * 13 + w + int(split))
    dmax = 96
    hp, wp = (h + 15) // 16 * 16, (w + 15) // 16 * 16
    x = rng.integers(-128, 128, (6, h, w), dtype=np.int8)
    dlow = (rng.random((hp // 16, wp // 16)) * (dmax / 16.0)).astype(np.float32)
    wt = (rng.standard_normal((32, 4, 3, 3)) / 4.0).astype(np.float32)
    b = rng.standard_normal(32).astype(np.float32)
    got = eng16.dbg_refin(dlow, x, dmax, wt, b, split)
    in4 = np.zeros((4, hp, wp), np.float32)
    in4[0] = oracle.upsample_bilinear(dlow, 16, 16.0) / np.float32(dmax)
    in4[1:, :h, :w] = x[:3].astype(np.float32) / 128.0
    ref = oracle.conv2d(in4, wt, b, 1, 1, 1)
    ref = np.where(ref > 0, ref, ref * np.float32(0.2))
    scale = max(1.0, np.abs(ref).max())
    if split:
        assert np.abs(got - ref).max() <= 3e-6 * scale + 1e-6      # 22-bit operands and a 22-bit output pair
    else:
        assert np.abs(got - q16(ref)).max() <= 1e-3 * scale        # one fp16 output rounding (+ ties)
        assert np.abs(got - ref).mean() < 2e-4 * scale
