"""Oracle network (C, fp32) vs the torch restatement: committed golden vectors and a
live torch run.  The network itself is parity-UNPINNED w.r.t. the reference (no
arithmetic, tests or vectors exist there — SURVEY.md §8(c)); these tests pin the
oracle to the published algorithm as restated twice, independently."""
import hashlib

import numpy as np
import pytest

from hobot_stereonet_amd import spec, synth, weights

CASES = [("c96x64_d48", 96, 64, 48, 3), ("c160x96_d96", 160, 96, 96, 4), ("c100x52_d32", 100, 52, 32, 5)]
TOL = 2e-4   # px; fp32 summation-order noise between the C loops and torch's kernels


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def test_weight_table_agrees_with_spec(oracle):
    assert oracle.weight_count() == spec.param_count() == 423586
    for name, (off, _) in spec.offsets().items():
        if name != "__total__":
            assert oracle.weight_offset(name) == off, name
    assert oracle.weight_offset("nope.w") == -1


def test_generators_have_not_drifted(golden_net, weights_blob):
    assert sha(weights_blob) == str(golden_net["weights_sha256"])
    for name, w, h, d, seed in CASES:
        assert sha(synth.model_input_i8(w, h, d, seed)) == str(golden_net[name + ".input_sha256"])


@pytest.mark.parametrize("name,w,h,d,seed", CASES)
def test_forward_matches_golden(oracle, golden_net, weights_blob, name, w, h, d, seed):
    x = synth.model_input_i8(w, h, d, seed)
    disp, raw, low = oracle.forward(weights_blob, x, d)
    assert np.abs(low - golden_net[name + ".disp_low"]).max() < 2e-5
    epe = np.abs(disp - golden_net[name + ".disp"]).mean()
    assert epe < TOL, epe
    assert np.abs(disp - golden_net[name + ".disp"]).max() < 20 * TOL
    # wire format: raw * scale * 16 * 12 reproduces disp to the int32 quantum, whatever D the model has
    q = 192 * spec.OUT_SCALE
    assert np.abs(raw.astype(np.float64) * q - disp).max() <= 0.52 * q + 2e-6
    assert raw.min() >= 0     # uint32 and int32 views agree (appendix B-5)


def test_multi_weight_table_and_file_format(oracle, weights_blob, weights_multi, tmp_path):
    """A hierarchical (`multi`) blob = the single blob + the towers of levels 1..3."""
    assert oracle.weight_count_levels(1) == spec.param_count(1)
    assert oracle.weight_count_levels(spec.MULTI_LEVELS) == spec.param_count(spec.MULTI_LEVELS) == 760933
    assert oracle.weight_count_levels(2) == -1
    for name, (off, _) in spec.offsets(spec.MULTI_LEVELS).items():
        if name != "__total__":
            assert oracle.weight_offset(name) == off, name
    assert np.array_equal(weights_multi[:weights_blob.size], weights_blob)
    p = str(tmp_path / "m.snw")
    weights.save_snw(p, weights_multi, 96, 64, 48)
    blob, meta = weights.load_snw(p)
    assert meta == {"width": 96, "height": 64, "dmax": 48, "levels": spec.MULTI_LEVELS} and np.array_equal(blob, weights_multi)
    with pytest.raises(ValueError):
        weights.save_snw(p, weights_multi[:-1], 96, 64, 48)
    assert abs(spec.flops_per_pair(1280, 720, 192, levels=4) / 1e9 - 295.56) < 0.01
    assert abs(spec.flops_per_pair(1242, 375, 256, levels=4) / 1e9 - 155.36) < 0.01


@pytest.mark.parametrize("name,w,h,d,seed", CASES)
def test_multi_forward_matches_golden(oracle, golden_multi, weights_multi, name, w, h, d, seed):
    assert sha(weights_multi) == str(golden_multi["weights_sha256"])
    x = synth.model_input_i8(w, h, d, seed)
    assert sha(x) == str(golden_multi[name + ".input_sha256"])
    disp, raw, low, maps = oracle.forward_levels(weights_multi, x, d)
    assert len(maps) == spec.MULTI_LEVELS - 1
    assert np.abs(low - golden_multi[name + ".disp_low"]).max() < 2e-5
    for k, m in enumerate(maps, start=1):                     # level k works at 1/2^k resolution
        gk = golden_multi[f"{name}.level{k}"]
        assert m.shape == gk.shape == (spec.ceil16(h) >> k, spec.ceil16(w) >> k)
        assert np.abs(m - gk).mean() < TOL and np.abs(m - gk).max() < 20 * TOL
    assert np.abs(disp - golden_multi[name + ".disp"]).mean() < TOL
    assert np.abs(disp - golden_multi[name + ".disp"]).max() < 20 * TOL
    q = 192 * spec.OUT_SCALE
    assert np.abs(raw.astype(np.float64) * q - disp).max() <= 0.52 * q + 2e-6 and raw.min() >= 0
    # the hierarchy is not a no-op: it differs from the single-scale result of the same leading weights
    single, _, _ = oracle.forward(weights_multi[:spec.param_count()], x, d)
    assert np.abs(single - disp).mean() > 0.05


def test_multi_ops_match_golden(oracle, golden_multi):
    g = golden_multi
    np.testing.assert_allclose(oracle.upsample_bilinear(g["op.up2.x"], 2, 2.0), g["op.up2.y"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(oracle.avgpool2(g["op.pool.x"]), g["op.pool.y"], rtol=1e-6, atol=1e-6)


def test_ops_match_golden(oracle, golden_net):
    g = golden_net
    x = g["op.conv2d.x"]
    for tag, s, p, dil in [("k3", 1, 1, 1), ("k3d4", 1, 4, 4), ("k5s2", 2, 2, 1)]:
        y = oracle.conv2d(x, g[f"op.conv2d.{tag}.w"], g[f"op.conv2d.{tag}.b"], s, p, dil)
        np.testing.assert_allclose(y, g[f"op.conv2d.{tag}.y"], rtol=1e-5, atol=2e-5)
    y3 = oracle.conv3d(g["op.conv3d.x"], g["op.conv3d.w"], g["op.conv3d.b"])
    np.testing.assert_allclose(y3, g["op.conv3d.y"], rtol=1e-5, atol=2e-5)
    np.testing.assert_allclose(oracle.upsample_bilinear(g["op.up.x"], 16, 16.0), g["op.up.y"], rtol=1e-6, atol=1e-5)
    np.testing.assert_allclose(oracle.soft_argmin(g["op.sam.x"]), g["op.sam.y"], rtol=1e-5, atol=1e-5)


def test_live_torch_agrees(oracle, weights_blob):
    import torch_ref
    w, h, d = 112, 80, 64
    x = synth.model_input_i8(w, h, d, 11)
    disp, _, low = oracle.forward(weights_blob, x, d)
    ref = torch_ref.forward(weights_blob, x, d)
    assert np.abs(low - ref["disp_low"]).max() < 2e-5
    assert np.abs(disp - ref["disp"]).mean() < TOL


def test_identical_eyes_known_answer(oracle, weights_blob):
    # the reference's only image fixtures are byte-identical L/R (config/image_left.jpg ==
    # image_right.jpg) => plane d=0 of the cost volume is exactly 0 and both feature maps are equal.
    w, h, d = 96, 64, 48
    x = synth.model_input_i8(w, h, d, 5).copy()
    x[3:] = x[:3]
    planes = np.zeros((3, h, w), np.float32)
    planes[:] = x[:3].astype(np.float32) / 128.0
    fl = oracle.features(weights_blob, planes)
    cv = oracle.cost_volume(fl, fl, d // 16)
    assert (cv[:, 0] == 0).all()
    assert (cv[:, 1, :, :1] == 0).all()


def test_bad_args_are_rejected(oracle, weights_blob):
    with pytest.raises(ValueError):
        oracle.forward(weights_blob, np.zeros((6, 16, 16), np.int8), 40)   # D not a multiple of 16


def test_snw_roundtrip(tmp_path, weights_blob):
    p = str(tmp_path / "m.snw")
    weights.save_snw(p, weights_blob, 1280, 720, 192)
    blob, meta = weights.load_snw(p)
    assert (blob == weights_blob).all() and meta == {"width": 1280, "height": 720, "dmax": 192, "levels": 1}
    with open(p, "r+b") as f:
        f.write(b"XXXX")
    with pytest.raises(ValueError):
        weights.load_snw(p)


def test_identical_eyes_reference_fixture_oracle(oracle, weights_blob):
    """The reference's only image fixture (config/image_left.jpg == image_right.jpg, committed as a 160x96 crop): the
    oracle's feature maps of the two eyes are equal, cost-volume plane 0 is exactly zero, the disparity is finite and
    non-negative (uint32 and int32 views of the wire tensor agree)."""
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "identical_eyes.npz"))
    bgr = g["bgr"]
    h, w = bgr.shape[:2]
    nv12 = oracle.bgr_to_nv12(bgr)
    x = oracle.preprocess_nv12(nv12, nv12, w, h)
    assert (x[:3] == x[3:]).all()
    planes = x[:3].astype(np.float32) / 128.0
    f = oracle.features(weights_blob, planes)
    cv = oracle.cost_volume(f, f, 6)
    assert (cv[:, 0] == 0).all()
    for dd in range(1, 6):
        assert (cv[:, dd, :, :dd] == 0).all()        # x < d: defined as zero
    disp, raw, low = oracle.forward(weights_blob, x, 96)
    assert np.isfinite(disp).all() and raw.min() >= 0


def test_baseline_config0_cpu_plumbing_960x540_d48(oracle, weights_blob):
    """BASELINE.json configs[0]: one 960x540 pair, D=48, CPU-float, no GPU — plumbing from an NV12 stereo frame to the
    consumer's depth image: reference pre-processing (split, YUV444, ^0x80) -> oracle network -> wire int32 -> the render
    node's view (uint32 * scale * 16 * 12, depth = f*B/disp/1000) -> colour image."""
    from hobot_stereonet_amd import render
    w, h, d = 960, 540, 48
    lt, rt = synth.stereo_pair_u8(w, h, d, 31)
    frame = np.random.default_rng(31).integers(0, 256, (h * 3 // 2, 2 * w), dtype=np.uint8)
    frame[:h, :w] = lt[0]
    frame[:h, w:] = rt[0]
    left, right = oracle.split_sbs_nv12(frame.ravel(), w, h)
    ten = oracle.preprocess_nv12(left, right, w, h)
    assert ten.shape == (6, h, w) and ten.dtype == np.int8
    assert np.array_equal(ten[0], (frame[:h, :w] ^ 0x80).view(np.int8))            # Quantize == b ^ 0x80 on the luma plane
    disp, raw, low = oracle.forward(weights_blob, ten, d)
    assert disp.shape == (h, w) and low.shape == (34, 60) and np.isfinite(disp).all() and raw.min() >= 0
    payload = raw.astype(np.int32).tobytes() + b"\xff\xd8jpeg-bytes-of-the-left-eye"
    raw_u32, rest = render.split_payload(payload, w, h)
    assert rest.startswith(b"\xff\xd8")
    dpx, depth = render.disparity_and_depth(raw_u32)
    assert np.abs(dpx - disp).max() <= 0.51 * 192 * spec.OUT_SCALE + 1e-6
    m = dpx > 1.0
    assert np.allclose(depth[m], 527.1931762695312 * 119.89382172 / dpx[m] / 1000.0, rtol=1e-5)
    rgb = render.colorize_depth(depth)
    assert rgb.shape == (h, w, 3) and rgb.dtype == np.uint8
