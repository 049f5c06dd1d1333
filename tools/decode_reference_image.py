#!/usr/bin/env python3
"""Decodes the render of the final scene that the reference's README shows (img/img2.png, 500x500) into the pixel
array the tests compare with: tests/golden/reference_final_scene_500x500.npz (`rgb` uint8 500x500x3, row 0 = top;
`png_sha256` of the file it came from).  Run in the build container, where /root/reference exists; the tests only
read the array.
usage: decode_reference_image.py [path to img2.png]"""
import hashlib
import sys
from pathlib import Path

import numpy as np
from PIL import Image

src = Path(sys.argv[1] if len(sys.argv) > 1 else "/root/reference/img/img2.png")
out = Path(__file__).resolve().parent.parent / "tests" / "golden" / "reference_final_scene_500x500.npz"
rgb = np.asarray(Image.open(src).convert("RGB"), dtype=np.uint8)
assert rgb.shape == (500, 500, 3), rgb.shape
np.savez_compressed(out, rgb=rgb, png_sha256=np.array(hashlib.sha256(src.read_bytes()).hexdigest()))
print(out, rgb.shape, out.stat().st_size, "bytes")
