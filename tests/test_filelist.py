"""File-list harness: the offline feeder `RunImglistFeedInfer` of the reference
(stereonet_infer/src/stereonet_node.cpp:820-976) — image readers, BGR->NV12 (preprocess.h:56-96), list error
behaviour and metrics on CPU; the C++ node feeder and its Python twin end to end against the oracle on the GPU."""
import ctypes as C
import json
import os
import subprocess

import numpy as np
import pytest

from hobot_stereonet_amd import filelist, images, spec, synth, weights

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
COMPAT = os.path.join(ROOT, "hobot_stereonet_amd", "csrc", "compat")
FILELIST_BIN = os.path.join(COMPAT, "build", "stereonet_filelist")


@pytest.fixture(scope="module")
def hostlib():
    from hobot_stereonet_amd import build
    build.build()
    san = os.environ.get("SN_SANITIZE") == "1"       # scripts/run_sanitized.sh: the ASan + UBSan build of the mirror
    subprocess.check_call(["make", "-C", COMPAT, "-s"] + (["asan"] if san else []))
    import torch  # noqa: F401  (before anything that links HIP: one HIP runtime per process, see api.load_library)
    lib = C.CDLL(os.path.join(COMPAT, "build", "asan" if san else "", "libhobot_stereonet_node.so"))
    vp, ci = C.c_void_p, C.c_int
    lib.snhost_bgr_to_nv12.argtypes = [vp, ci, ci, vp]
    lib.snhost_read_image_bgr.argtypes = [C.c_char_p, C.POINTER(ci), C.POINTER(ci), vp, C.c_long]
    lib.snhost_pfm_roundtrip.argtypes = [C.c_char_p, vp, ci, ci, vp]
    return lib


def host_read(hostlib, path):
    w, h = C.c_int(), C.c_int()
    if hostlib.snhost_read_image_bgr(path.encode(), C.byref(w), C.byref(h), None, 0) != 0:
        return None
    out = np.empty((h.value, w.value, 3), np.uint8)
    assert hostlib.snhost_read_image_bgr(path.encode(), C.byref(w), C.byref(h), out.ctypes.data, out.size) == 0
    return out


# ---- BGR -> NV12 ----------------------------------------------------------------------------------------------
def test_bgr_to_nv12_bt601_known_answers(oracle, hostlib):
    """BT.601 studio-range code values of the primaries (ITU-R BT.601 table): the pin of the restated OpenCV
    arithmetic (OpenCV itself is not in the image: parity with it is otherwise unpinned)."""
    cases = {  # (B, G, R) -> (Y, U, V)
        (255, 255, 255): (235, 128, 128), (0, 0, 0): (16, 128, 128), (0, 0, 255): (82, 90, 240),
        (0, 255, 0): (145, 54, 34), (255, 0, 0): (41, 240, 110), (128, 128, 128): (126, 128, 128),
    }
    for bgr, yuv in cases.items():
        img = np.empty((2, 2, 3), np.uint8)
        img[:] = bgr
        for impl in (images.bgr_to_nv12, oracle.bgr_to_nv12):
            nv = impl(img)
            assert tuple(nv[:4]) == (yuv[0],) * 4 and (nv[4], nv[5]) == yuv[1:], (bgr, nv)
        out = np.empty(6, np.uint8)
        assert hostlib.snhost_bgr_to_nv12(img.ctypes.data, 2, 2, out.ctypes.data) == 0
        assert tuple(out) == (yuv[0],) * 4 + yuv[1:]


def test_bgr_to_nv12_three_implementations_agree(oracle, hostlib):
    rng = np.random.default_rng(5)
    for (w, h) in ((2, 2), (16, 8), (34, 18), (160, 96)):
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        a = images.bgr_to_nv12(img)
        b = oracle.bgr_to_nv12(img)
        c = np.empty(w * h * 3 // 2, np.uint8)
        assert hostlib.snhost_bgr_to_nv12(img.ctypes.data, w, h, c.ctypes.data) == 0
        assert (a == b).all() and (a == c).all()
        # layout: Y plane, then h/2 rows of interleaved U,V taken from the even/even pixel
        assert a.size == w * h * 3 // 2
        uv = a[w * h:].reshape(h // 2, w // 2, 2)
        one = images.bgr_to_nv12(np.broadcast_to(img[0:1, 0:1], (2, 2, 3)).copy())
        assert tuple(uv[0, 0]) == (one[4], one[5])
    odd = np.zeros((4, 5, 3), np.uint8)        # the reference rejects odd sizes (preprocess.h:60-63)
    with pytest.raises(ValueError):
        images.bgr_to_nv12(odd)
    with pytest.raises(ValueError):
        oracle.bgr_to_nv12(odd)
    assert hostlib.snhost_bgr_to_nv12(odd.ctypes.data, 5, 4, np.empty(64, np.uint8).ctypes.data) == -1


# ---- image files ------------------------------------------------------------------------------------------------
def test_png_ppm_pfm_round_trips(hostlib, tmp_path):
    rng = np.random.default_rng(11)
    rgb = rng.integers(0, 256, (18, 26, 3), dtype=np.uint8)
    gray = rng.integers(0, 256, (18, 26), dtype=np.uint8)
    g16 = rng.integers(0, 65536, (18, 26), dtype=np.uint16)
    p = str(tmp_path / "a.png")
    images.write_png(p, rgb)
    assert (images.read_png(p) == rgb).all()
    assert (images.imread_bgr(p) == rgb[..., ::-1]).all()
    assert (host_read(hostlib, p) == rgb[..., ::-1]).all()
    images.write_png(p, gray)
    assert (images.read_png(p) == gray).all()
    assert (host_read(hostlib, p) == gray[..., None]).all()
    images.write_png(p, g16)
    assert (images.read_png(p) == g16).all()
    assert host_read(hostlib, p) is None          # 16-bit is ground truth, not camera input
    q = str(tmp_path / "a.ppm")
    images.write_ppm(q, rgb)
    assert (images.imread_bgr(q) == rgb[..., ::-1]).all()
    assert (host_read(hostlib, q) == rgb[..., ::-1]).all()
    # PFM: python <-> C++ (bottom-up rows, little endian)
    d = rng.standard_normal((7, 9)).astype(np.float32) * 50
    f = str(tmp_path / "d.pfm")
    images.write_pfm(f, d)
    assert (images.read_pfm(f) == d).all()
    back = np.empty_like(d)
    assert hostlib.snhost_pfm_roundtrip(str(tmp_path / "e.pfm").encode(), d.ctypes.data, 9, 7, back.ctypes.data) == 0
    assert (back == d).all() and (images.read_pfm(str(tmp_path / "e.pfm")) == d).all()
    gt, valid = images.read_disparity(f)
    assert (valid == (d > 0)).all()
    images.write_png(p, (np.abs(d) * 256).astype(np.uint16))
    gt, valid = images.read_disparity(p)           # KITTI convention
    assert np.allclose(gt, (np.abs(d) * 256).astype(np.uint16) / 256.0) and (valid == (gt > 0)).all()
    assert host_read(hostlib, str(tmp_path / "missing.png")) is None


def test_png_filters_and_bmp_written_by_pil(hostlib, tmp_path):
    """Files from an independent encoder: adaptive PNG filters (Sub/Up/Average/Paeth), RGBA, palette, BMP."""
    Image = pytest.importorskip("PIL.Image")
    w, h = 40, 24
    left, _ = synth.stereo_pair_u8(w, h, 16, 3)            # smooth content -> the encoder picks non-trivial filters
    rgb = np.ascontiguousarray(left.transpose(1, 2, 0))
    p = str(tmp_path / "pil.png")
    Image.fromarray(rgb).save(p, optimize=True)
    assert (images.imread_bgr(p) == rgb[..., ::-1]).all()
    assert (host_read(hostlib, p) == rgb[..., ::-1]).all()
    rgba = np.dstack([rgb, np.full((h, w), 200, np.uint8)])
    Image.fromarray(rgba).save(p)
    assert (images.imread_bgr(p) == rgb[..., ::-1]).all()          # alpha dropped, as IMREAD_COLOR does
    assert (host_read(hostlib, p) == rgb[..., ::-1]).all()
    pal = Image.fromarray(rgb).quantize(16)
    pal.save(p)
    want = np.asarray(pal.convert("RGB"))[..., ::-1]
    assert (images.imread_bgr(p) == want).all()
    assert (host_read(hostlib, p) == want).all()
    b = str(tmp_path / "pil.bmp")
    Image.fromarray(rgb).save(b)
    assert (host_read(hostlib, b) == rgb[..., ::-1]).all()


# ---- lists and metrics ----------------------------------------------------------------------------------------------
def test_list_error_behaviour(tmp_path):
    a, b = str(tmp_path / "a.ppm"), str(tmp_path / "b.ppm")
    for p in (a, b):
        images.write_ppm(p, np.zeros((4, 4, 3), np.uint8))
    ll, rl = str(tmp_path / "l.list"), str(tmp_path / "r.list")
    open(ll, "w").write(f"{a}\n{b}\n")
    open(rl, "w").write(f"{a}\n")
    with pytest.raises(filelist.FileListError, match="Imgs size error"):           # stereonet_node.cpp:881-887
        filelist.read_pair_lists(ll, rl)
    with pytest.raises(filelist.FileListError, match="Open file failed"):          # :833-838
        filelist.read_pair_lists(str(tmp_path / "none.list"), rl)
    open(rl, "w").write(f"{a}\n{tmp_path}/ghost.png\n")
    with pytest.raises(filelist.FileListError, match="File is not exist"):         # :843-847
        filelist.read_pair_lists(ll, rl)
    open(rl, "w").write(f"{a}\r\n{b}\n")                                            # CRLF lists are tolerated
    assert filelist.read_pair_lists(ll, rl) == ([a, b], [a, b])


def test_metrics_by_hand():
    gt = np.array([[10.0, 20.0, 100.0, 0.0]])
    pred = np.array([[10.5, 24.0, 103.5, 50.0]])
    valid = gt > 0
    assert abs(filelist.epe(pred, gt, valid) - (0.5 + 4.0 + 3.5) / 3) < 1e-12
    assert abs(filelist.bad_px(pred, gt, 1.0, valid) - 2 / 3) < 1e-12
    assert abs(filelist.bad_px(pred, gt, 3.0, valid) - 2 / 3) < 1e-12
    # D1: >3 px AND >5 %: 24 vs 20 (4 px, 20 %) is bad; 103.5 vs 100 (3.5 px, 3.5 %) is not
    assert abs(filelist.d1(pred, gt, valid) - 1 / 3) < 1e-12
    s = filelist.score(pred, gt, valid, dmax=50.0)      # the 100-px pixel is outside the search range
    assert s["valid_px"] == 2 and abs(s["epe"] - 2.25) < 1e-12


def test_filelist_binary_fails_loudly_without_gpu(hostlib, tmp_path):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    m = str(tmp_path / "m.snw")
    weights.save_snw(m, weights.synthetic(0), 96, 64, 48)
    for n in ("l.list", "r.list"):
        open(str(tmp_path / n), "w").write("")
    r = subprocess.run([FILELIST_BIN, m, str(tmp_path / "l.list"), str(tmp_path / "r.list"), str(tmp_path)],
                       capture_output=True, text=True)
    assert r.returncode == 3 and "Node init fail!" in r.stderr
    # the Python twin: lists are checked first, then the engine refuses to come up without a device
    rc = filelist.main(["--model", m, "--left", str(tmp_path / "nope.list"), "--right", str(tmp_path / "r.list")])
    assert rc == 5
    from hobot_stereonet_amd import api
    with pytest.raises(api.StereoNetError):
        filelist.main(["--model", m, "--left", str(tmp_path / "l.list"), "--right", str(tmp_path / "r.list")])


# ---- end to end on the GPU ----------------------------------------------------------------------------------------
def _write_dataset(tmp_path, w, h, d, n):
    lefts, rights, gts = [], [], []
    for i in range(n):
        l, r = synth.stereo_pair_u8(w, h, d, 20 + i)
        lp, rp, gp = (str(tmp_path / f"{k}{i}.{e}") for k, e in (("left", "png"), ("right", "ppm"), ("gt", "pfm")))
        images.write_png(lp, np.ascontiguousarray(l.transpose(1, 2, 0)))          # planes taken as R,G,B
        images.write_ppm(rp, np.ascontiguousarray(r.transpose(1, 2, 0)))
        images.write_pfm(gp, synth.disparity_field(w, h, d))
        lefts.append(lp), rights.append(rp), gts.append(gp)
    names = {}
    for k, v in (("left", lefts), ("right", rights), ("gt", gts)):
        names[k] = str(tmp_path / f"{k}.list")
        open(names[k], "w").write("\n".join(v) + "\n")
    return names, lefts, rights


@pytest.mark.gpu
def test_filelist_cpp_and_python_match_the_oracle(hostlib, oracle, weights_blob, model_factory, tmp_path):
    from hobot_stereonet_amd import api
    w, h, d, n = 160, 96, 96, 3
    m = model_factory(w, h, d)
    names, lefts, rights = _write_dataset(tmp_path, w, h, d, n)
    out_cpp, out_py = tmp_path / "cpp", tmp_path / "py"
    out_cpp.mkdir()
    env = dict(os.environ, STEREONET_PRECISION="fp32", STEREONET_FEED_PAUSE_MS="0")
    r = subprocess.run([FILELIST_BIN, m, names["left"], names["right"], str(out_cpp)], capture_output=True, text=True,
                       env=env, timeout=300)
    assert r.returncode == 0, r.stderr
    lines = [l for l in r.stdout.splitlines() if l.startswith("frame_id=")]
    assert [l.split()[0] for l in lines] == [f"frame_id={i}" for i in range(n)]          # frame_id = list index (:925)
    assert f"fed={n} received={n}" in r.stdout
    with api.StereoNetHIP(m, precision=api.PREC_FP32) as eng:
        recs = filelist.run_imglist(eng, names["left"], names["right"], str(out_py), names["gt"])
    assert len(recs) == n
    for i in range(n):
        # the oracle on the same files: imread -> BGRToNv12 -> CvtNV12Data2Tensors -> network
        eyes = [oracle.bgr_to_nv12(images.imread_bgr(p)) for p in (lefts[i], rights[i])]
        ten = oracle.preprocess_nv12(eyes[0], eyes[1], w, h)
        odisp, oraw, _ = oracle.forward(weights_blob, ten, d)
        raw_cpp = np.fromfile(str(out_cpp / f"{i}.raw.bin"), np.int32).reshape(h, w)
        raw_py = np.fromfile(str(out_py / f"{i}.raw.bin"), np.int32).reshape(h, w)
        assert (raw_cpp == raw_py).all() and (raw_py == recs[i]["raw"]).all()      # host and device pre-processing agree
        assert np.abs(recs[i]["disp"] - odisp).mean() < 1e-3                       # EPE vs the oracle, px
        pfm = images.read_pfm(str(out_cpp / f"{i}.disp.pfm"))          # the harness dequantises with the render node's 16*12
        assert np.abs(pfm - odisp).mean() < 1e-3
        assert (images.read_pfm(str(out_py / f"{i}.disp.pfm")) == recs[i]["disp"]).all()
        assert os.path.getsize(str(out_cpp / f"{i}.jpg")) > 200
        assert images.read_pnm(str(out_py / f"{i}.depth.ppm")).shape == (h, w, 3)
        mt = recs[i]["metrics"]
        assert mt["valid_px"] > 0 and np.isfinite(mt["epe"]) and 0.0 <= mt["d1"] <= 1.0
    # the command line prints one JSON summary
    r = subprocess.run(["python", "-m", "hobot_stereonet_amd.filelist", "--model", m, "--left", names["left"], "--right",
                        names["right"], "--gt", names["gt"], "--precision", "fp32"], capture_output=True, text=True,
                       cwd=ROOT, timeout=300)
    assert r.returncode == 0, r.stderr
    s = json.loads(r.stdout.strip().splitlines()[-1])
    assert s["frames"] == n and abs(s["epe"] - np.mean([x["metrics"]["epe"] for x in recs])) < 1e-6


@pytest.mark.gpu
def test_filelist_error_paths_on_the_node(hostlib, model_factory, tmp_path):
    w, h, d = 160, 96, 96
    m = model_factory(w, h, d)
    names, lefts, rights = _write_dataset(tmp_path, w, h, d, 2)
    env = dict(os.environ, STEREONET_FEED_PAUSE_MS="0", SN_LOG_LEVEL="3")
    out = str(tmp_path)
    short = str(tmp_path / "short.list")
    open(short, "w").write(lefts[0] + "\n")
    r = subprocess.run([FILELIST_BIN, m, names["left"], short, out], capture_output=True, text=True, env=env, timeout=120)
    assert r.returncode == 5 and "Imgs size error" in r.stderr and "fed=0" in r.stdout
    r = subprocess.run([FILELIST_BIN, m, str(tmp_path / "none.list"), short, out], capture_output=True, text=True, env=env,
                       timeout=120)
    assert r.returncode == 5 and "Open file failed" in r.stderr
    small = str(tmp_path / "small.ppm")
    images.write_ppm(small, np.zeros((h // 2, w // 2, 3), np.uint8))
    bad = str(tmp_path / "bad.list")
    open(bad, "w").write(f"{lefts[0]}\n{small}\n")
    r = subprocess.run([FILELIST_BIN, m, names["left"], bad, out], capture_output=True, text=True, env=env, timeout=120)
    assert r.returncode == 5 and "BGRToNv12 Fail" in r.stderr and "fed=1 received=1" in r.stdout   # frame 0 went through
