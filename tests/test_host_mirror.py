"""Host C++ mirror of the reference node (hobot_stereonet_amd/csrc/compat): PreProcess / Parse / JPEG pure
functions through ctypes on CPU; StereonetNode end to end through the in-process rclcpp stand-in on the GPU."""
import ctypes as C
import io
import os
import subprocess

import numpy as np
import pytest

from hobot_stereonet_amd import spec, synth, weights

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
COMPAT = os.path.join(ROOT, "hobot_stereonet_amd", "csrc", "compat")


@pytest.fixture(scope="module")
def hostlib():
    from hobot_stereonet_amd import build
    build.build()
    san = os.environ.get("SN_SANITIZE") == "1"       # scripts/run_sanitized.sh: the ASan + UBSan build of the mirror
    subprocess.check_call(["make", "-C", COMPAT, "-s"] + (["asan"] if san else []))
    import torch  # noqa: F401  (before anything that links HIP: one HIP runtime per process, see api.load_library)
    lib = C.CDLL(os.path.join(COMPAT, "build", "asan" if san else "", "libhobot_stereonet_node.so"))
    vp, ci = C.c_void_p, C.c_int
    lib.snhost_yuv420_to_yuv444.argtypes = [vp, vp, ci, ci]
    lib.snhost_quantize_byte.argtypes = [ci]
    lib.snhost_jpeg_nv12.restype = C.c_long
    lib.snhost_jpeg_nv12.argtypes = [vp, ci, ci, ci, ci, vp, C.c_long]
    lib.snhost_jpeg_nv12_sliced.restype = C.c_long
    lib.snhost_jpeg_nv12_sliced.argtypes = [vp, ci, ci, ci, ci, ci, vp, C.c_long]
    lib.snhost_jpeg_pool.restype = C.c_long
    lib.snhost_jpeg_pool.argtypes = [vp, ci, ci, ci, ci, ci, ci, ci, vp, C.c_long]
    lib.snhost_jpeg_nv12_reference.restype = C.c_long
    lib.snhost_jpeg_nv12_reference.argtypes = [vp, ci, ci, ci, ci, vp, C.c_long]
    lib.snhost_parse.argtypes = [vp, ci, ci, C.c_float, vp, vp]
    return lib


def test_host_yuv444_and_quantize_match_reference_vectors(hostlib, golden_pre):
    for c in ("ramp8x4", "rand32x16", "rand64x36", "rand48x20"):
        w, h = map(int, golden_pre[c + ".wh"])
        src = np.ascontiguousarray(golden_pre[c + ".nv12"])
        out = np.empty(3 * w * h, np.uint8)
        hostlib.snhost_yuv420_to_yuv444(src.ctypes.data, out.ctypes.data, w, h)
        assert (out == golden_pre[c + ".yuv444"]).all(), c
    table = np.array([hostlib.snhost_quantize_byte(b) for b in range(256)], np.int8)
    assert (table == golden_pre["quant_table"]).all()


def test_host_parse_matches_oracle_and_reference_formula(hostlib, oracle):
    w, h = 16, 4
    raw = np.arange(1, w * h + 1, dtype=np.int32) * 3000
    depth = np.empty(w * h, np.float32)
    disp = np.empty(w * h, np.float32)
    scale = C.c_float(spec.OUT_SCALE)
    assert hostlib.snhost_parse(raw.ctypes.data, w, h, scale, depth.ctypes.data, disp.ctypes.data) == 0
    odisp, odepth = oracle.dequant_depth(raw, spec.OUT_SCALE, 192.0)
    np.testing.assert_allclose(disp, odisp, rtol=1e-6)
    np.testing.assert_allclose(depth, odepth, rtol=1e-5)
    # parser.cpp:84-86 by hand: int32 200000 -> 0.632 m
    one = np.array([200000], np.int32)
    assert hostlib.snhost_parse(one.ctypes.data, 1, 1, scale, depth.ctypes.data, disp.ctypes.data) == 0
    assert abs(depth[0] - 0.632) < 1e-3 and abs(disp[0] - 100.01) < 1e-2


def test_host_jpeg_decodes_to_the_source_image(hostlib):
    from PIL import Image
    w, h = 96, 64
    left, _ = synth.stereo_pair_u8(w, h, 48, 2)
    y = np.clip(left[0], 48, 208)          # keep RGB in gamut so the PIL round trip does not clip
    uv = np.full((h // 2, w), 128, np.uint8)      # neutral chroma
    uv[:, 0::2] = 90
    uv[:, 1::2] = 170
    nv12 = np.concatenate([y.ravel(), uv.ravel()])
    buf = np.empty(w * h * 3, np.uint8)
    n = hostlib.snhost_jpeg_nv12(nv12.ctypes.data, w, h, w, 95, buf.ctypes.data, buf.size)
    assert n > 100
    img = Image.open(io.BytesIO(buf[:n].tobytes()))
    assert img.size == (w, h) and img.mode == "RGB"
    ycc = np.asarray(img.convert("YCbCr"), np.float32)
    assert np.abs(ycc[..., 0] - y).mean() < 2.0            # luma survives quality-95 quantisation
    assert abs(ycc[..., 1].mean() - 90) < 3 and abs(ycc[..., 2].mean() - 170) < 3
    # side-by-side source (pitch = 2w): the left half only
    sbs = np.zeros((h * 3 // 2, 2 * w), np.uint8)
    sbs[:h, :w] = y
    sbs[h:, :w] = uv
    n2 = hostlib.snhost_jpeg_nv12(sbs.ctypes.data, w, h, 2 * w, 95, buf.ctypes.data, buf.size)
    img2 = np.asarray(Image.open(io.BytesIO(buf[:n2].tobytes())).convert("YCbCr"), np.float32)
    assert np.abs(img2[..., 0] - y).mean() < 2.0


@pytest.mark.parametrize("w,h,q", [(1280, 720, 95), (96, 64, 95), (70, 50, 75), (1242, 374, 50), (34, 18, 100)])
def test_fast_jpeg_matches_the_exact_dct_encoder(hostlib, w, h, q):
    """The node's encoder (AAN DCT with the scale factors folded into the quantisers, branch-free bit writer, stuffing in a
    second pass) against the exact separable-DCT encoder of round 3: same headers and size class, decoded images within
    50 dB of each other (coefficients may differ by one step where a value sits on a rounding boundary), and both as close
    to the source as each other.  Sizes cover blocks cut by the right / bottom edge and frames of odd MCU counts."""
    from PIL import Image
    W = w + (w & 1)
    fr = synth.sbs_nv12_frame(W, h, 32, 5).reshape(h * 3 // 2, 2 * W)
    buf, ref = np.empty(W * h * 4 + 8192, np.uint8), np.empty(W * h * 4 + 8192, np.uint8)
    n = hostlib.snhost_jpeg_nv12(fr.ctypes.data, W, h, 2 * W, q, buf.ctypes.data, buf.size)
    m = hostlib.snhost_jpeg_nv12_reference(fr.ctypes.data, W, h, 2 * W, q, ref.ctypes.data, ref.size)
    assert n > 600 and m > 600 and abs(n - m) < 0.01 * m + 64
    assert bytes(buf[:623]) == bytes(ref[:623])                        # SOI .. SOS: identical tables and frame header
    assert bytes(buf[n - 2:n]) == b"\xff\xd9"
    a = np.asarray(Image.open(io.BytesIO(buf[:n].tobytes())).convert("YCbCr"), np.float64)
    b = np.asarray(Image.open(io.BytesIO(ref[:m].tobytes())).convert("YCbCr"), np.float64)
    assert a.shape == (h, W, 3)
    mse = np.mean((a - b) ** 2)
    assert mse == 0 or 10 * np.log10(255.0 ** 2 / mse) >= 50.0
    src_y = fr[:h, :W].astype(np.float64)
    assert abs(np.mean((a[..., 0] - src_y) ** 2) - np.mean((b[..., 0] - src_y) ** 2)) < 0.05


@pytest.mark.parametrize("w,h,per", [(1280, 720, 6), (96, 64, 1), (70, 50, 3), (64, 48, 2), (1242, 374, 5)])
def test_sliced_jpeg_decodes_to_the_same_image(hostlib, w, h, per):
    """Restart-interval form (what the node's encoder threads produce, one slice per task): DRI + RSTm markers in the
    stream, and a decoded image identical to the single-scan stream's, pixel for pixel."""
    from PIL import Image
    W = w + (w & 1)
    fr = synth.sbs_nv12_frame(W, h, 32, 9).reshape(h * 3 // 2, 2 * W)
    one, sl = np.empty(W * h * 4 + 8192, np.uint8), np.empty(W * h * 4 + 8192, np.uint8)
    n = hostlib.snhost_jpeg_nv12(fr.ctypes.data, W, h, 2 * W, 95, one.ctypes.data, one.size)
    m = hostlib.snhost_jpeg_nv12_sliced(fr.ctypes.data, W, h, 2 * W, 95, per, sl.ctypes.data, sl.size)
    rows = (h + 15) // 16
    nsl = (rows + per - 1) // per
    assert m > 0 and (bytes(sl[:m]).count(b"\xff\xdd\x00\x04") == 1) == (nsl > 1)
    if nsl > 1:
        assert sum(bytes(sl[:m]).count(bytes([0xFF, 0xD0 + k])) for k in range(8)) >= nsl - 1
    a = np.asarray(Image.open(io.BytesIO(one[:n].tobytes())))
    b = np.asarray(Image.open(io.BytesIO(sl[:m].tobytes())))
    assert a.shape == (h, W, 3) and (a == b).all()


@pytest.mark.parametrize("w,h,threads,slices,frames", [(640, 360, 4, 8, 12), (96, 64, 3, 4, 40), (70, 50, 2, 9, 25), (64, 48, 1, 1, 5)])
def test_encoder_threads_assemble_the_sliced_stream(hostlib, w, h, threads, slices, frames):
    """The node's encoder-thread path (JpegPool + SubmitSlicedJpeg: one task per slice, the last one to finish assembles)
    without the node: many frames in flight together, every assembled stream identical to EncodeNv12ToJpegSliced's and
    decodable.  Also runs under ASan / UBSan (scripts/run_sanitized.sh)."""
    from PIL import Image
    W = w + (w & 1)
    fr = synth.sbs_nv12_frame(W, h, 32, 13).reshape(h * 3 // 2, 2 * W)
    buf = np.empty(W * h * 4 + 8192, np.uint8)
    n = hostlib.snhost_jpeg_pool(fr.ctypes.data, W, h, 2 * W, 90, threads, slices, frames, buf.ctypes.data, buf.size)
    assert n > 600, n
    img = Image.open(io.BytesIO(buf[:n].tobytes()))
    assert img.size == (W, h)
    assert np.abs(np.asarray(img.convert("YCbCr"), np.float32)[..., 0] - fr[:h, :W]).mean() < 6.0


def test_encoder_threads_under_thread_sanitizer():
    """`make tsan`: JpegPool + SubmitSlicedJpeg built with -fsanitize=thread, three feeder threads x 20 frames x 1 / 4 / 11
    slices on six encoder threads; every stream equals the sequential encoder's and ThreadSanitizer reports nothing
    (a report makes the binary exit non-zero)."""
    r = subprocess.run(["make", "-C", COMPAT, "-s", "tsan"], capture_output=True, text=True, timeout=600)
    if r.returncode != 0 and ("cannot find -ltsan" in r.stderr or "libtsan" in r.stderr):
        pytest.skip("ThreadSanitizer runtime not installed")
    assert r.returncode == 0, (r.stdout + r.stderr)[-2000:]
    assert r.stdout.count("0 mismatching streams of 60") == 3 and "WARNING: ThreadSanitizer" not in r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("publish", [1, 0])
def test_node_level_throughput_is_recorded(hostlib, weights_blob, tmp_path, publish):
    """a-1 at node level (stereonet_node.cpp:657-818 + 980-1089): frames/s through StereonetNode — FeedImg, the left-eye
    JPEG on the encoder threads, asynchronous Run with 4 requests in flight, PostProcess, the published message — on a
    seeded 1280x720 stereo frame.  The figure is recorded (printed, and in gpurun_out/ when that exists); the floor
    asserted here only catches a return to the 25 frames/s of an encoder on the executor thread."""
    import json
    w, h, d = 1280, 720, 192
    m = str(tmp_path / "m.snw")
    weights.save_snw(m, weights_blob, w, h, d)
    synth.sbs_nv12_frame(w, h, d, 21).tofile(str(tmp_path / "s.bin"))
    env = dict(os.environ, STEREONET_PUB_OUTPUT=str(publish), SN_LOG_LEVEL="3")
    r = subprocess.run([os.path.join(COMPAT, "build", "node_harness"), "--bench", m, str(tmp_path / "s.bin"), str(w), str(h),
                        "400"], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    res = json.loads(line)
    print(line)
    out_dir = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out_dir):
        with open(os.path.join(out_dir, f"node_bench_publish{publish}.json"), "w") as f:
            f.write(line + "\n")
    assert res["frames"] == 400 and res["publish"] is bool(publish)
    assert res["frames_per_s"] > 100.0
    if publish:
        assert res["payload_bytes_per_frame"] > 4 * w * h + 1000


def test_node_init_fails_loudly_without_gpu(hostlib, tmp_path):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    m = str(tmp_path / "m.snw")
    weights.save_snw(m, weights.synthetic(0), 96, 64, 48)
    sbs = np.zeros((64 * 3 // 2) * 192, np.uint8)
    sbs.tofile(str(tmp_path / "s.bin"))
    r = subprocess.run([os.path.join(COMPAT, "build", "node_harness"), m, str(tmp_path / "s.bin"), "96", "64", "1",
                        str(tmp_path / "o")], capture_output=True, text=True)
    assert r.returncode == 3 and "Node init fail!" in r.stderr       # stereonet_node.cpp:44-48
    # missing model file: SetNodePara's access() check (stereonet_node.cpp:131-134)
    r = subprocess.run([os.path.join(COMPAT, "build", "node_harness"), str(tmp_path / "nope.snw"),
                        str(tmp_path / "s.bin"), "96", "64", "1", str(tmp_path / "o")], capture_output=True, text=True)
    assert r.returncode == 3 and "File is not exist" in r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("ingest,multi", [("nv12", False), ("tensor", False), ("nv12", True)])
def test_stereonet_node_end_to_end(hostlib, oracle, weights_blob, weights_multi, tmp_path, ingest, multi):
    """hbmem NV12 frame in -> /stereonet_node_output message out, payload = int32 tensor || JPEG(left).
    ingest = nv12: the node hands the raw message payload to the backend (split + CvtNV12Data2Tensors on the GPU, the
    default); tensor: the reference's host steps and Run() on the int8 tensor.  Both must publish the same bytes.
    multi: model_file holds the hierarchical-refinement network — nothing else changes for the node."""
    from PIL import Image
    w, h, d = 96, 64, 48
    m = str(tmp_path / "m.snw")
    if multi:
        weights_blob = weights_multi
    weights.save_snw(m, weights_blob, w, h, d)
    # band-limited luma (white noise is pathological for JPEG), random chroma bytes
    lt, rt = synth.stereo_pair_u8(w, h, d, 8)
    frame = np.random.default_rng(8).integers(0, 256, (h * 3 // 2, 2 * w), dtype=np.uint8)
    frame[:h, :w] = lt[0]
    frame[:h, w:] = rt[0]
    sbs = frame.ravel()
    sbs.tofile(str(tmp_path / "s.bin"))
    nframes = 6          # > task_num: exercises the 4 in-flight slots
    env = dict(os.environ, STEREONET_PRECISION="fp32", STEREONET_INGEST=ingest)
    r = subprocess.run([os.path.join(COMPAT, "build", "node_harness"), m, str(tmp_path / "s.bin"), str(w), str(h),
                        str(nframes), str(tmp_path / "o")], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr
    lines = [l for l in r.stdout.splitlines() if l.startswith("frame_id=")]
    assert len(lines) == nframes and f"received={nframes}" in r.stdout
    assert [l.split()[0] for l in lines] == [f"frame_id={100 + i}" for i in range(nframes)]     # in order
    assert all("encoding=jpeg" in l and f"height={h} width={w}" in l and "stamp=7." in l for l in lines)
    # bad-encoding / bad-geometry frames were dropped, not published (stereonet_node.cpp:672-690)
    left, right = oracle.split_sbs_nv12(sbs, w, h)
    ten = oracle.preprocess_nv12(left, right, w, h)
    odisp, oraw, _ = oracle.forward(weights_blob, ten, d)
    for i in range(nframes):
        payload = np.fromfile(str(tmp_path / f"o.{i}.msg"), dtype=np.uint8)
        raw = payload[:w * h * 4].view(np.uint32).reshape(h, w)        # the render node's view (uint32)
        disp = raw.astype(np.float64) * spec.OUT_SCALE * 16 * 12      # the literal factor, whatever D (here 96) is
        assert np.abs(disp - odisp).mean() < 1e-3
        assert np.abs(raw.astype(np.int64) - oraw).max() <= 2
        jpg = Image.open(io.BytesIO(payload[w * h * 4:].tobytes()))
        assert jpg.size == (w, h)
        # the render node's twin consumes the message as is (it hard-codes the 16*12 of the reference model)
        from hobot_stereonet_amd import render
        rdisp, rdepth, joint = render.render(payload.tobytes(), w, h)
        assert joint.shape == (2 * h, w, 3)
        assert np.abs(rdisp - odisp).mean() < 1e-3
        y = np.asarray(jpg.convert("YCbCr"), np.float32)[..., 0]
        assert np.abs(y - left[:w * h].reshape(h, w)).mean() < 6.0


def test_cpp_render_twin_matches_python_twin(hostlib):
    """Row f-3 in C++ (compat/src/render.cpp): payload split (uint32 view), dequantisation, depth, convertScaleAbs(9),
    JET table, BGR-as-RGB stacking — byte for byte the numpy twin (hobot_stereonet_amd/render.py), incl. raw = 0 -> inf
    depth -> 255 and the table values remembered from OpenCV's colormap.cpp."""
    from hobot_stereonet_amd import render
    vp = C.c_void_p
    hostlib.snhost_render.argtypes = [vp, C.c_long, C.c_int, C.c_int, vp, vp, vp, vp, vp]
    hostlib.snhost_jet_lut.argtypes = [vp]
    lut = np.empty((256, 3), np.uint8)
    hostlib.snhost_jet_lut(lut.ctypes.data)
    assert (lut == render.jet_lut()).all()
    w, h = 40, 12
    rng = np.random.default_rng(3)
    raw = rng.integers(0, 400000, (h, w)).astype(np.int32)
    raw[0, :5] = [0, 1, 200000, 2 ** 31 - 1, 10]
    payload = np.frombuffer(raw.tobytes() + b"\xff\xd8JPEG", np.uint8).copy()
    left = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    disp = np.empty((h, w), np.float64)
    depth = np.empty((h, w), np.float64)
    color = np.empty((h, w, 3), np.uint8)
    joint = np.empty((2 * h, w, 3), np.uint8)
    off = hostlib.snhost_render(payload.ctypes.data, payload.size, w, h, disp.ctypes.data, depth.ctypes.data,
                                color.ctypes.data, left.ctypes.data, joint.ctypes.data)
    assert off == w * h * 4 and bytes(payload[off:]) == b"\xff\xd8JPEG"
    pr, _ = render.split_payload(payload.tobytes(), w, h)
    pdisp, pdepth = render.disparity_and_depth(pr)
    assert (disp == pdisp).all() and (depth == pdepth).all()
    assert (color == render.colorize_depth(pdepth)).all()
    assert (joint[:h] == left).all() and (joint[h:] == color).all()
    assert hostlib.snhost_render(payload.ctypes.data, w * h * 4 - 1, w, h, None, None, None, None, None) == -1
