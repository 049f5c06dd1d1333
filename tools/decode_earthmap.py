#!/usr/bin/env python3
"""Decode the reference's earth texture into the raw RGB8 asset the renderer loads.

The reference opens textures/earthmap.jpg with the `image` crate at run time
(texture.rs:91-98).  Neither side of this build links a JPEG decoder, so the
PIL (libjpeg-turbo) decode is stored once as 1024x512 RGB8, row 0 = image top,
and used by both the HIP path and the CPU oracle.

Usage: python tools/decode_earthmap.py /root/reference/textures/earthmap.jpg
"""
import hashlib
import sys
from pathlib import Path

from PIL import Image

src = Path(sys.argv[1])
dst = Path(__file__).resolve().parent.parent / "raytracing_amd" / "assets" / "earthmap_1024x512.rgb"
img = Image.open(src).convert("RGB")
assert img.size == (1024, 512), img.size
raw = img.tobytes()
dst.write_bytes(raw)
print(dst, len(raw), hashlib.sha256(raw).hexdigest())
