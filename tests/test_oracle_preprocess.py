"""Oracle pre-processing vs vectors produced by the reference's own functions
(tests/golden/make_preprocess_golden.py) — pins SURVEY.md §8 rows a-2, a-3, a-4."""
import numpy as np

from hobot_stereonet_amd import synth

CASES = ["ramp8x4", "rand32x16", "rand64x36", "rand48x20"]


def test_quantize_table_matches_reference(oracle, golden_pre):
    # preprocess.cpp:1038 + :1131-1136 evaluated for every byte value by the reference code
    ref = golden_pre["quant_table"]
    mine = np.array([oracle.quantize((float(b) - 128.0) / 128.0) for b in range(256)], np.int8)
    assert (mine == ref).all()
    # ... and is exactly b ^ 0x80 reinterpreted as int8 (what the HIP path computes)
    assert (ref.view(np.uint8) == (np.arange(256, dtype=np.uint8) ^ 0x80)).all()


def test_yuv420_to_yuv444_matches_reference(oracle, golden_pre):
    for c in CASES:
        w, h = golden_pre[c + ".wh"]
        got = oracle.yuv420_to_yuv444(golden_pre[c + ".nv12"], int(w), int(h))
        assert (got.ravel() == golden_pre[c + ".yuv444"]).all(), c


def test_ramp_known_answer(oracle):
    # 8x4 ramp -> "U" = bytes 32..39 replicated 2x2, "V" = bytes 40..47
    w, h = 8, 4
    img = np.arange(w * h * 3 // 2, dtype=np.uint8)
    out = oracle.yuv420_to_yuv444(img, w, h)
    assert (out[0].ravel() == img[:32]).all()
    exp_u = np.repeat(np.repeat(np.arange(32, 40, dtype=np.uint8).reshape(2, 4), 2, 0), 2, 1)
    exp_v = np.repeat(np.repeat(np.arange(40, 48, dtype=np.uint8).reshape(2, 4), 2, 0), 2, 1)
    assert (out[1] == exp_u).all() and (out[2] == exp_v).all()


def test_preprocess_is_yuv444_xor_0x80(oracle, golden_pre):
    # CvtNV12Data2Tensors (preprocess.cpp:975-1056): L planes then R planes, each byte quantised
    c = "rand64x36"
    w, h = map(int, golden_pre[c + ".wh"])
    left = golden_pre[c + ".nv12"]
    right = synth.random_nv12(w, h, 99)
    got = oracle.preprocess_nv12(left, right, w, h)
    exp_l = golden_pre[c + ".yuv444"].reshape(3, h, w) ^ np.uint8(0x80)
    assert got.dtype == np.int8 and got.shape == (6, h, w)
    assert (got[:3].view(np.uint8) == exp_l).all()
    assert (got[3:].view(np.uint8) == (oracle.yuv420_to_yuv444(right, w, h) ^ np.uint8(0x80))).all()


def test_split_side_by_side(oracle):
    # stereonet_node.cpp:705-738: every 2w-byte source row -> first w bytes left, last w bytes right,
    # for the h luma rows and then the h/2 chroma rows.
    w, h = 16, 8
    sbs = np.random.default_rng(3).integers(0, 256, (h * 3 // 2, 2 * w), dtype=np.uint8)
    left, right = oracle.split_sbs_nv12(sbs.ravel(), w, h)
    assert (left.reshape(-1, w) == sbs[:, :w]).all()
    assert (right.reshape(-1, w) == sbs[:, w:]).all()


def test_dequant_depth_known_answer(oracle):
    # int32 200000 -> dis 0.520888 -> 100.01 px -> 0.632 m (parser.cpp:70-71,84-86)
    disp, depth = oracle.dequant_depth(np.array([200000], np.int32), 2.60443857769133e-6, 192.0)
    assert abs(disp[0] - 100.0104) < 1e-3
    assert abs(depth[0] - 527.1931762695312 * 119.89382172 / 100.0104 / 1000.0) < 1e-4
    # zero disparity -> IEEE inf, not an exception (appendix B-10)
    _, depth0 = oracle.dequant_depth(np.array([0], np.int32), 2.60443857769133e-6, 192.0)
    assert np.isinf(depth0[0])
