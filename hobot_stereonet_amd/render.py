"""Twin of the reference's render node:
``stereonet_render_tools/hobot_stereonet_render/publisher_member_function.py:45-163`` —
``MinimalPublisher.listener_callback`` takes the ``/stereonet_node_output`` payload (int32 tensor ‖ JPEG of
the left eye), turns it into metric depth, colour-maps it and stacks it under the left image for ``/image_jpeg``.

The reference is Python + cv2 + rclpy; cv2 and rclpy are not installed here, so this twin is numpy + PIL and
re-states the three cv2 calls it needs:
  * ``cv2.convertScaleAbs(src, alpha)``  = saturate_cast<uint8>(|src*alpha|), round-half-even, NaN -> 0, inf -> 255
  * ``cv2.applyColorMap(_, COLORMAP_JET)`` = 256-entry LUT r,g,b(x = i/255) = clip(1.5 - |4x - {3,2,1}|, 0, 1)
    (checked against the two table values remembered from OpenCV's colormap.cpp: r[96] = 0.00588235…,
    r[97] = 0.02156862…); PARITY UNPINNED — no cv2 in this image to diff the table or the JPEG bytes against.
  * ``cv2.imdecode`` / ``cv2.imencode('.jpg')`` -> PIL (default cv2 quality 95).
Quirks reproduced on purpose: the payload is viewed as **uint32**; the colour map
goes through a BGR-as-RGB PIL round trip, so its red and blue channels end up swapped in the published image.
"""
from __future__ import annotations

import io
from typing import Tuple

import numpy as np

SCALE = 0.00000260443857769133      # publisher_member_function.py:29
FOCAL = 527.1931762695312           # :30
BASELINE = 119.89382172             # :31


def jet_lut() -> np.ndarray:
    """COLORMAP_JET as a (256, 3) uint8 table in cv2's BGR channel order."""
    x = np.arange(256, dtype=np.float64) / 255.0
    r = np.clip(1.5 - np.abs(4.0 * x - 3.0), 0.0, 1.0)
    g = np.clip(1.5 - np.abs(4.0 * x - 2.0), 0.0, 1.0)
    b = np.clip(1.5 - np.abs(4.0 * x - 1.0), 0.0, 1.0)
    return np.rint(np.stack([b, g, r], axis=1) * 255.0).astype(np.uint8)


def convert_scale_abs(src: np.ndarray, alpha: float) -> np.ndarray:
    with np.errstate(invalid="ignore", over="ignore"):
        v = np.abs(np.asarray(src, np.float64) * alpha)
        out = np.where(np.isnan(v), 0.0, np.minimum(np.rint(v), 255.0))
    return out.astype(np.uint8)


def split_payload(data: bytes, w: int, h: int) -> Tuple[np.ndarray, bytes]:
    """:57-66 — first w*h*4 bytes = model output viewed as uint32, rest = JPEG of the left eye."""
    n = w * h * 4
    if len(data) < n:
        raise ValueError("payload shorter than the model output tensor")
    raw = np.frombuffer(data[:n], dtype=np.uint32).reshape(h, w)
    return raw, bytes(data[n:])


def disparity_and_depth(raw_u32: np.ndarray, dmax_factor: float = 16 * 12) -> Tuple[np.ndarray, np.ndarray]:
    """:72-81 — image_pre = raw * scale * 16 * 12 (px); Z = f*B/image_pre/1000 (m); 0 disparity -> inf."""
    disp = raw_u32.astype(np.float64) * SCALE * dmax_factor
    with np.errstate(divide="ignore"):
        depth = FOCAL * BASELINE / disp / 1000.0
    return disp, depth


def colorize_depth(depth: np.ndarray, alpha: float = 9.0) -> np.ndarray:
    """:82 — applyColorMap(convertScaleAbs(Z, alpha=9), COLORMAP_JET) -> (h, w, 3) uint8 in BGR order."""
    return jet_lut()[convert_scale_abs(depth, alpha)]


def render(data: bytes, w: int, h: int):
    """-> (disp float64 (h,w), depth float64 (h,w), joint uint8 (2h, w, 3) RGB as finally published)."""
    from PIL import Image
    raw, jpeg = split_payload(data, w, h)
    disp, depth = disparity_and_depth(raw)
    color_bgr = colorize_depth(depth)
    left_rgb = np.asarray(Image.open(io.BytesIO(jpeg)).convert("RGB"))     # :94-101: imdecode (BGR) + r/b swap = RGB
    # :108,121-137: the BGR colour map is handed to PIL as if it were RGB, pasted under the (true RGB) left image
    # and the whole canvas is channel-reversed for cv2.imencode -> in true colours the colour map's R and B swap
    joint = np.concatenate([left_rgb, color_bgr], axis=0)
    return disp, depth, joint


def encode_jpeg(joint_rgb: np.ndarray, quality: int = 95) -> bytes:
    """:159 — cv2.imencode('.jpg') of the stacked image (what goes out on /image_jpeg)."""
    from PIL import Image
    buf = io.BytesIO()
    Image.fromarray(joint_rgb, "RGB").save(buf, format="JPEG", quality=quality)
    return buf.getvalue()
