#!/usr/bin/env python3
"""Builds tests/golden/identical_eyes.npz from the reference's only image fixture.

The reference ships stereonet_infer/config/image_left.jpg and image_right.jpg — byte-identical files (SURVEY.md §2 row
8) — as the input of its (disabled) offline feeder.  Identical eyes are a known-answer input: the two feature maps are
equal, so cost-volume plane d = 0 is exactly zero wherever it is defined.  This script (run in the build container,
where /root/reference exists) decodes the JPEG with PIL, takes a 1280x720-proportioned centre crop, box-downscales it
to 160x96 and stores the BGR pixels (46 KB) plus the sha256 of the two source files.  Only that data is committed; the
test derives NV12 / the model tensor from it with the oracle."""
import hashlib
import os

import numpy as np
from PIL import Image

REF = "/root/reference/stereonet_infer/config"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "identical_eyes.npz")


def main():
    paths = [os.path.join(REF, n) for n in ("image_left.jpg", "image_right.jpg")]
    shas = [hashlib.sha256(open(p, "rb").read()).hexdigest() for p in paths]
    assert shas[0] == shas[1], "the reference's two fixture images are expected to be byte-identical"
    im = Image.open(paths[0]).convert("RGB")
    W, H = im.size
    cw, ch = (W, W * 9 // 16) if W * 9 // 16 <= H else (H * 16 // 9, H)
    left, top = (W - cw) // 2, (H - ch) // 2
    small = im.crop((left, top, left + cw, top + ch)).resize((160, 96), Image.BOX)
    bgr = np.asarray(small, np.uint8)[:, :, ::-1].copy()
    np.savez_compressed(OUT, bgr=bgr, source_sha256=np.array(shas[0]), source_size=np.array([W, H]))
    print(OUT, bgr.shape, shas[0])


if __name__ == "__main__":
    main()
