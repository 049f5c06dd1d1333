"""Progressive accumulation with checkpoint / resume.

`Frame` is the binding of the C ABI's mirt_frame_* entry points: the sums stay on the GPU between passes, a
pass adds into them on the device, and the library reads and writes the checkpoints.  `Progressive` is the same
bookkeeping on the host over any object with a render() method (it reads and writes the same checkpoint format).

The reference renders a frame in one go (renderer.rs:43-92).  A frame here is a sum over samples whose random
streams depend only on (pixel, sample index, seed), so it can be rendered in passes over sample ranges, stopped,
written to disk and continued later -- on any number of GPUs -- and ends up the same image (up to the order of
the f32 additions) as an uninterrupted render.

    acc = Progressive(scene, 1920, 1080, seed=7, aspect=16 / 9)
    acc.render(64); acc.save("frame.ckpt")            # ... later, possibly another process:
    acc = Progressive.load(scene, "frame.ckpt"); acc.render(960); rgb = acc.image()
"""
import ctypes as C
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
                           "depth": self.depth, "flags": 0, "started": int(self.samples > 0),
                           "aspect": self.aspect}).encode()
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
            acc = cls(scene, h["width"], h["height"], seed=h["seed"], depth=h["depth"], aspect=h.get("aspect", 0.0))
            raw = f.read()
        if len(raw) != acc.sums.nbytes:
            raise ValueError(f"{path}: truncated checkpoint")
        acc.sums = np.frombuffer(raw, "<f4").reshape(acc.sums.shape).astype(np.float32)
        acc.samples = int(h["samples"])
        return acc


class Frame:
    """A frame kept on the GPU by the library (mirt_frame_*): passes, previews, checkpoints."""

    def __init__(self, scene, width=None, height=None, seed=7, depth=None, aspect=0.0, flags=0, _handle=None):
        import raytracing_amd as R
        self.R, self.L, self.scene = R, R.mirt(), scene
        self.ctx = scene.L.rt_example_context(scene.h)
        self.dev_scene = scene.L.rt_example_scene(scene.h)
        self.seed, self.flags, self.aspect = int(seed), int(flags), float(aspect)
        self.depth = scene.depth if depth is None else int(depth)
        if _handle is None:
            self.width, self.height = int(width), int(height)
            h = C.c_void_p()
            self._check(self.L.mirt_frame_create(self.ctx, self.width, self.height, C.byref(h)))
            self.h = h
        else:
            self.h, self.width, self.height = _handle, int(width), int(height)

    def _check(self, rc):
        if rc != 0:
            raise self.R.MirtError(f"status {rc}: {self.L.mirt_last_error(self.ctx).decode()}")

    @property
    def samples(self):
        return int(self.L.mirt_frame_samples(self.h))

    def render(self, spp, pass_size=None):
        """Adds `spp` more samples per pixel, in passes of `pass_size` (default: one pass)."""
        cam, bg = self.scene.camera(self.aspect or self.width / self.height)
        left = int(spp)
        step = int(pass_size) if pass_size else left
        stats = None
        while left > 0:
            n = min(step, left)
            p = self.R.RenderParams(self.width, self.height, n, 0, self.depth, (C.c_float * 3)(*bg), self.seed, 0,
                                    self.flags, 0, 0)
            st = self.R.Stats()
            self._check(self.L.mirt_frame_render(self.ctx, self.dev_scene, C.byref(cam), C.byref(p), self.h, C.byref(st)))
            stats = st.as_dict()
            left -= n
        return stats

    def read(self):
        """(rgb24 as Renderer::present would show it now, f32 sums)."""
        rgb = np.zeros((self.height, self.width, 3), np.uint8)
        sums = np.zeros((self.height, self.width, 3), np.float32)
        self._check(self.L.mirt_frame_read(self.ctx, self.h, rgb.ctypes.data, sums.ctypes.data))
        return rgb, sums

    def save(self, path):
        self._check(self.L.mirt_frame_save(self.ctx, self.h, str(path).encode()))

    @classmethod
    def load(cls, scene, path, aspect=0.0):
        import raytracing_amd as R
        L = R.mirt()
        ctx = scene.L.rt_example_context(scene.h)
        h = C.c_void_p()
        rc = L.mirt_frame_load(ctx, str(path).encode(), C.byref(h))
        if rc != 0:
            raise R.MirtError(f"status {rc}: {L.mirt_last_error(ctx).decode()}")
        with open(path, "rb") as f:
            f.read(8)
            (n,) = struct.unpack("<I", f.read(4))
            head = json.loads(f.read(n))
        return cls(scene, head["width"], head["height"], seed=head["seed"], depth=head["depth"], aspect=aspect,
                   flags=head.get("flags", 0), _handle=h)

    def close(self):
        if getattr(self, "h", None):
            self.L.mirt_frame_destroy(self.ctx, self.h)
            self.h = None
