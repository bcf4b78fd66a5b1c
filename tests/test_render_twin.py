"""Render-node twin (hobot_stereonet_amd/render.py) — formula KATs for SURVEY.md §8 row f-3.  cv2 is absent, so the
colour table and JPEG bytes are parity-unpinned; what is pinned is every formula the reference spells out
(publisher_member_function.py:57-82) and the remembered OpenCV table values."""
import io

import numpy as np

from hobot_stereonet_amd import render, spec


def test_payload_split_and_dequant():
    w, h = 8, 4
    raw = (np.arange(w * h, dtype=np.int32) * 7919 + 1000).reshape(h, w)
    payload = raw.tobytes() + b"JPEGBYTES"
    got, jpeg = render.split_payload(payload, w, h)
    assert got.dtype == np.uint32 and (got == raw.astype(np.uint32)).all() and jpeg == b"JPEGBYTES"
    disp, depth = render.disparity_and_depth(got)
    np.testing.assert_allclose(disp, raw * spec.OUT_SCALE * 192, rtol=1e-12)
    np.testing.assert_allclose(depth, 527.1931762695312 * 119.89382172 / disp / 1000.0, rtol=1e-12)
    # SURVEY KAT: 200000 -> 100.01 px -> 0.632 m ; 0 -> inf
    d, z = render.disparity_and_depth(np.array([[200000, 0]], np.uint32))
    assert abs(d[0, 0] - 100.0104) < 1e-3 and abs(z[0, 0] - 0.632) < 1e-3 and np.isinf(z[0, 1])


def test_convert_scale_abs_semantics():
    src = np.array([0.0, 0.05, 0.0556, 1.0, 28.3, 28.4, -3.0, np.inf, np.nan])
    got = render.convert_scale_abs(src, 9)
    assert got.tolist() == [0, 0, 1, 9, 255, 255, 27, 255, 0]          # |x*9| rounded, saturated; NaN -> 0
    assert render.convert_scale_abs(np.array([0.5 / 9, 1.5 / 9, 2.5 / 9]), 9).tolist() == [0, 2, 2]   # half-even


def test_jet_table():
    lut = render.jet_lut()
    assert lut.shape == (256, 3) and lut.dtype == np.uint8
    assert tuple(lut[0]) == (128, 0, 0)            # BGR: dark blue
    assert tuple(lut[255]) == (0, 0, 128)          # dark red
    assert tuple(lut[128])[1] == 255               # green plateau in the middle
    # OpenCV colormap.cpp Jet r[] first non-zero entries: 0.00588235294117645, 0.02156862745098032
    x = np.arange(256) / 255.0
    r = np.clip(1.5 - np.abs(4 * x - 3), 0, 1)
    assert abs(r[96] - 0.00588235294117645) < 1e-12 and abs(r[97] - 0.02156862745098032) < 1e-12 and r[95] == 0


def test_render_stacks_left_over_swapped_colormap():
    from PIL import Image
    w, h = 64, 32
    left = np.zeros((h, w, 3), np.uint8)
    left[..., 0] = 200                                   # a red left image
    buf = io.BytesIO()
    Image.fromarray(left, "RGB").save(buf, format="JPEG", quality=95)
    raw = np.full((h, w), 200000, np.int32)              # 0.632 m -> gray level round(0.632*9) = 6
    raw[:, w // 2:] = 0                                  # zero disparity -> inf -> 255
    disp, depth, joint = render.render(raw.tobytes() + buf.getvalue(), w, h)
    assert joint.shape == (2 * h, w, 3)
    assert abs(int(joint[:h, :, 0].mean()) - 200) < 4 and joint[:h, :, 2].mean() < 6      # top: the left eye, true colours
    lut = render.jet_lut()
    assert (joint[h:, : w // 2] == lut[6]).all()          # bottom: BGR table entries land in RGB slots (R/B swapped)
    assert (joint[h:, w // 2:] == lut[255]).all() and tuple(lut[255]) == (0, 0, 128)       # "inf" shows as blue
    out = render.encode_jpeg(joint)
    assert Image.open(io.BytesIO(out)).size == (w, 2 * h)
