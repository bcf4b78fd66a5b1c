"""Image files of the offline (file-list) path: what `cv::imread` / `Tools::BGRToNv12` do for the reference's
`RunImglistFeedInfer` (stereonet_infer/src/stereonet_node.cpp:901-921, include/preprocess.h:56-96), plus the
disparity formats stereo datasets use for ground truth (PFM: SceneFlow / Middlebury; 16-bit PNG / 256: KITTI).

No OpenCV / PIL dependency: PNG through zlib, PPM/PGM and PFM by hand.  Host-side code; the network itself only
ever runs through libstereonet_hip.so.
"""
import re
import struct
import zlib
from typing import Tuple

import numpy as np

_PNG_SIG = b"\x89PNG\r\n\x1a\n"


# ---- PNG ------------------------------------------------------------------------------------------------
def _png_unfilter(raw: np.ndarray, h: int, stride: int, bpp: int) -> np.ndarray:
    rows = raw.reshape(h, stride + 1)
    out = np.zeros((h, stride), np.uint8)
    prev = np.zeros(stride, np.int32)
    for r in range(h):
        ft = int(rows[r, 0])
        cur = rows[r, 1:].astype(np.int32)
        if ft == 0:
            rec = cur
        elif ft == 2:
            rec = (cur + prev) & 255
        elif ft == 1:      # Sub: independent running sums per byte lane of the pixel
            rec = cur.copy()
            for lane in range(bpp):
                rec[lane::bpp] = np.cumsum(cur[lane::bpp]) & 255
        elif ft in (3, 4):   # Average / Paeth depend on the reconstructed left neighbour: sequential
            rec = np.zeros(stride, np.int32)
            p = prev
            for i in range(stride):
                a = rec[i - bpp] if i >= bpp else 0
                b = p[i]
                if ft == 3:
                    pred = (a + b) >> 1
                else:
                    c = p[i - bpp] if i >= bpp else 0
                    pa, pb, pc = abs(b - c), abs(a - c), abs(a + b - 2 * c)
                    pred = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
                rec[i] = (cur[i] + pred) & 255
        else:
            raise ValueError(f"png: bad filter type {ft}")
        out[r] = rec
        prev = rec
    return out


def read_png(path: str) -> np.ndarray:
    """-> (h, w) or (h, w, c) array, uint8 or uint16 (big-endian samples converted), channels in file order (RGB[A])."""
    data = open(path, "rb").read()
    if data[:8] != _PNG_SIG:
        raise ValueError(f"{path}: not a PNG file")
    pos, idat, plte, hdr = 8, [], None, None
    while pos + 12 <= len(data):
        n, tag = struct.unpack(">I4s", data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + n]
        if tag == b"IHDR":
            hdr = struct.unpack(">IIBBBBB", body[:13])
        elif tag == b"PLTE":
            plte = np.frombuffer(body, np.uint8).reshape(-1, 3)
        elif tag == b"IDAT":
            idat.append(body)
        elif tag == b"IEND":
            break
        pos += 12 + n
    if hdr is None:
        raise ValueError(f"{path}: no IHDR")
    w, h, depth, ctype, _, _, interlace = hdr
    if interlace:
        raise ValueError(f"{path}: interlaced PNG is not supported")
    if depth not in (1, 2, 4, 8, 16) or (depth < 8 and ctype not in (0, 3)):
        raise ValueError(f"{path}: bit depth {depth} is not supported")
    spp = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[ctype]
    bpp = max(1, spp * depth // 8)
    stride = (w * spp * depth + 7) // 8
    raw = np.frombuffer(zlib.decompress(b"".join(idat)), np.uint8)
    if raw.size != h * (stride + 1):
        raise ValueError(f"{path}: truncated image data")
    px = _png_unfilter(raw, h, stride, bpp)
    if depth == 16:
        img = px.reshape(h, w, spp, 2).astype(np.uint16)
        img = (img[..., 0] << 8) | img[..., 1]
    elif depth < 8:       # packed samples, most significant bits first; gray levels are stretched to 0..255
        bits = np.unpackbits(px, axis=1)[:, :w * depth].reshape(h, w, depth)
        img = (bits * (1 << np.arange(depth - 1, -1, -1))).sum(axis=2).astype(np.uint8)[..., None]
        if ctype == 0:
            img = (img.astype(np.uint16) * 255 // ((1 << depth) - 1)).astype(np.uint8)
    else:
        img = px.reshape(h, w, spp)
    if ctype == 3:
        if plte is None:
            raise ValueError(f"{path}: palette image without PLTE")
        img = plte[img[..., 0]]
    return img[..., 0] if img.shape[-1] == 1 else img


def write_png(path: str, img: np.ndarray) -> None:
    """uint8 (h,w) / (h,w,3) / (h,w,4) or uint16 (h,w); filter type 0, one IDAT."""
    img = np.asarray(img)
    if img.ndim == 2:
        img = img[..., None]
    h, w, c = img.shape
    ctype = {1: 0, 3: 2, 4: 6}[c]
    if img.dtype == np.uint16:
        depth, body = 16, img.astype(">u2").tobytes()
    elif img.dtype == np.uint8:
        depth, body = 8, img.tobytes()
    else:
        raise ValueError("write_png: uint8 or uint16 expected")
    stride = len(body) // h
    rows = b"".join(b"\x00" + body[r * stride:(r + 1) * stride] for r in range(h))

    def chunk(tag, payload):
        return struct.pack(">I", len(payload)) + tag + payload + struct.pack(">I", zlib.crc32(tag + payload) & 0xFFFFFFFF)

    with open(path, "wb") as f:
        f.write(_PNG_SIG + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, ctype, 0, 0, 0)) +
                chunk(b"IDAT", zlib.compress(rows, 6)) + chunk(b"IEND", b""))


# ---- PPM / PGM ------------------------------------------------------------------------------------------
def read_pnm(path: str) -> np.ndarray:
    data = open(path, "rb").read()
    m = re.match(rb"P([56])\s+(?:#[^\n]*\n\s*)*(\d+)\s+(?:#[^\n]*\n\s*)*(\d+)\s+(?:#[^\n]*\n\s*)*(\d+)\s", data)
    if not m or int(m.group(4)) != 255:
        raise ValueError(f"{path}: binary PPM/PGM with maxval 255 expected")
    w, h, c = int(m.group(2)), int(m.group(3)), 3 if m.group(1) == b"6" else 1
    px = np.frombuffer(data, np.uint8, count=w * h * c, offset=m.end())
    return px.reshape(h, w, c) if c == 3 else px.reshape(h, w)


def write_ppm(path: str, rgb: np.ndarray) -> None:
    rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
    h, w = rgb.shape[:2]
    with open(path, "wb") as f:
        f.write((f"P6\n{w} {h}\n255\n" if rgb.ndim == 3 else f"P5\n{w} {h}\n255\n").encode() + rgb.tobytes())


# ---- PFM ------------------------------------------------------------------------------------------------
def read_pfm(path: str) -> np.ndarray:
    """Single-channel ('Pf') or colour ('PF') PFM -> float32 (h,w[,3]), top row first."""
    with open(path, "rb") as f:
        magic = f.readline().strip()
        if magic not in (b"Pf", b"PF"):
            raise ValueError(f"{path}: not a PFM file")
        dims = f.readline().split()
        while len(dims) < 2:
            dims += f.readline().split()
        w, h = int(dims[0]), int(dims[1])
        scale = float(f.readline().strip())
        c = 3 if magic == b"PF" else 1
        a = np.frombuffer(f.read(4 * w * h * c), "<f4" if scale < 0 else ">f4")
    if a.size != w * h * c:
        raise ValueError(f"{path}: truncated")
    a = a.reshape(h, w, c)[::-1].astype(np.float32)
    return a[..., 0] if c == 1 else a


def write_pfm(path: str, a: np.ndarray) -> None:
    a = np.asarray(a, dtype="<f4")
    h, w = a.shape
    with open(path, "wb") as f:
        f.write(f"Pf\n{w} {h}\n-1.0\n".encode() + np.ascontiguousarray(a[::-1]).tobytes())


# ---- imread / BGR -> NV12 ----------------------------------------------------------------------------------
def imread_bgr(path: str) -> np.ndarray:
    """The `cv::imread(path, cv::IMREAD_COLOR)` role: any supported file -> (h, w, 3) uint8, B,G,R order."""
    with open(path, "rb") as f:
        head = f.read(8)
    if head == _PNG_SIG:
        img = read_png(path)
        if img.dtype != np.uint8:
            raise ValueError(f"{path}: 8-bit colour input expected")
    elif head[:2] in (b"P5", b"P6"):
        img = read_pnm(path)
    else:
        raise ValueError(f"{path}: unsupported image format")
    if img.ndim == 2:
        img = np.repeat(img[..., None], 3, axis=2)
    elif img.shape[2] == 2:
        img = np.repeat(img[..., :1], 3, axis=2)
    return np.ascontiguousarray(img[..., 2::-1] if img.shape[2] >= 3 else img)


def bgr_to_nv12(bgr: np.ndarray) -> np.ndarray:
    """Tools::BGRToNv12 (preprocess.h:56-96): BT.601 studio range in Q20 with round-half-up, chroma sampled at
    the top-left pixel of each 2x2 block, U/V interleaved.  Returns the flat NV12 buffer (w*h*3/2 bytes)."""
    bgr = np.asarray(bgr, dtype=np.uint8)
    h, w = bgr.shape[:2]
    if h % 2 or w % 2:
        raise ValueError("input img height and width must aligned by 2!")
    b, g, r = (bgr[..., i].astype(np.int64) for i in range(3))
    half, q = 1 << 19, 20
    y = (269484 * r + 528482 * g + 102760 * b + half + (16 << q)) >> q
    r0, g0, b0 = r[::2, ::2], g[::2, ::2], b[::2, ::2]
    u = (-155188 * r0 - 305135 * g0 + 460324 * b0 + half + (128 << q)) >> q
    v = (460324 * r0 - 385875 * g0 - 74448 * b0 + half + (128 << q)) >> q
    uv = np.stack([u, v], axis=-1).reshape(h // 2, w)
    return np.clip(np.concatenate([y, uv], axis=0), 0, 255).astype(np.uint8).reshape(-1)


def sbs_from_eyes(left_nv12: np.ndarray, right_nv12: np.ndarray, w: int, h: int) -> np.ndarray:
    """Two w x h NV12 images -> the side-by-side 2w x h NV12 frame the live node receives."""
    rows = h * 3 // 2
    return np.ascontiguousarray(np.concatenate([left_nv12.reshape(rows, w), right_nv12.reshape(rows, w)], axis=1))


def read_disparity(path: str) -> Tuple[np.ndarray, np.ndarray]:
    """Ground-truth disparity -> (float32 map, valid mask).  PFM: finite positive values are valid;
    16-bit PNG: value/256 with 0 = invalid (the KITTI convention)."""
    if path.lower().endswith(".pfm"):
        d = read_pfm(path)
        return d, np.isfinite(d) & (d > 0)
    img = read_png(path)
    if img.dtype == np.uint16:
        d = img.astype(np.float32) / 256.0
    else:
        d = img.astype(np.float32)
    return d, img > 0
