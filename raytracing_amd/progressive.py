"""Progressive accumulation with checkpoint / resume on top of the C ABI.

The reference renders a frame in one go (renderer.rs:43-92).  A frame here is a sum over samples whose random
streams depend only on (pixel, sample index, seed), so it can be rendered in passes over sample ranges, stopped,
written to disk and continued later -- on any number of GPUs -- and ends up the same image (up to the order of
the f32 additions) as an uninterrupted render.

    acc = Progressive(scene, 1920, 1080, seed=7, aspect=16 / 9)
    acc.render(64); acc.save("frame.ckpt")            # ... later, possibly another process:
    acc = Progressive.load(scene, "frame.ckpt"); acc.render(960); rgb = acc.image()
"""
import json
import struct

import numpy as np

MAGIC = b"MIRTACC1"


def encode_rgb24(sums, spp):
    """renderer.rs:241-252 on the host: (255.999 * clamp(sqrt(c / spp), 0, 0.999)) as u8, NaN -> 0."""
    with np.errstate(invalid="ignore", divide="ignore"):
        v = np.sqrt(sums.astype(np.float32) / np.float32(spp))
        v = np.where(v < 0, np.float32(0), np.where(v > np.float32(0.999), np.float32(0.999), v))  # clamp keeps NaN
        v = np.float32(255.999) * v
    out = np.zeros(v.shape, np.uint8)
    ok = np.isfinite(v) & (v > 0)
    out[ok] = np.minimum(v[ok], 255).astype(np.uint8)
    return out


class Progressive:
    """Sample sums of one frame plus what is needed to continue it."""

    def __init__(self, scene, width, height, seed=7, depth=None, aspect=0.0):
        self.scene = scene
        self.width, self.height, self.seed, self.aspect = int(width), int(height), int(seed), float(aspect)
        self.depth = scene.depth if depth is None else int(depth)
        self.samples = 0                                              # samples [0, samples) are in `sums`
        self.sums = np.zeros((self.height, self.width, 3), np.float32)

    def render(self, spp, pass_size=None, **kw):
        """Adds samples [samples, samples + spp) in passes of `pass_size` (default: one pass)."""
        left = int(spp)
        step = int(pass_size) if pass_size else left
        while left > 0:
            n = min(step, left)
            _, s, _ = self.scene.render(self.width, self.height, n, depth=self.depth, aspect=self.aspect,
                                        seed=self.seed, first_sample=self.samples, **kw)
            self.sums += s
            self.samples += n
            left -= n
        return self

    def image(self):
        if self.samples == 0:
            return np.zeros((self.height, self.width, 3), np.uint8)
        return encode_rgb24(self.sums, self.samples)

    # -- checkpoint: magic, header length, JSON header, raw f32 sums ------------------------------------
    def save(self, path):
        head = json.dumps({"width": self.width, "height": self.height, "samples": self.samples, "seed": self.seed,
                           "depth": self.depth, "aspect": self.aspect}).encode()
        with open(path, "wb") as f:
            f.write(MAGIC + struct.pack("<I", len(head)) + head)
            f.write(np.ascontiguousarray(self.sums, "<f4").tobytes())

    @classmethod
    def load(cls, scene, path):
        with open(path, "rb") as f:
            if f.read(8) != MAGIC:
                raise ValueError(f"{path}: not a checkpoint of this library")
            (n,) = struct.unpack("<I", f.read(4))
            h = json.loads(f.read(n))
            acc = cls(scene, h["width"], h["height"], seed=h["seed"], depth=h["depth"], aspect=h["aspect"])
            raw = f.read()
        if len(raw) != acc.sums.nbytes:
            raise ValueError(f"{path}: truncated checkpoint")
        acc.sums = np.frombuffer(raw, "<f4").reshape(acc.sums.shape).astype(np.float32)
        acc.samples = int(h["samples"])
        return acc
