"""raytracing_amd: MI355X (gfx950) path-tracing backend for the ShadowCurse/raytracing renderer.

The product is two shared libraries built in-tree:
  libmirt.so          gfx950 kernels + the C ABI of include/mirt.h
  librt_examples.so   the C++ host mirror of the reference API with its example scenes

This module is a thin ctypes binding over them (plumbing for tests and bench.py).  It
never falls back to a CPU implementation: if the libraries are missing or no GPU is
present the calls raise.
"""
import ctypes as C
import os
import subprocess
from pathlib import Path

import numpy as np

PKG_DIR = Path(__file__).resolve().parent
ROOT = PKG_DIR.parent
ASSET_DIR = PKG_DIR / "assets"
# MIRT_LIBMIRT names another build of the library (kernel A/B comparisons on one GPU box)
LIBMIRT = Path(os.environ["MIRT_LIBMIRT"]) if os.environ.get("MIRT_LIBMIRT") else PKG_DIR / "libmirt.so"
LIBEXAMPLES = PKG_DIR / "librt_examples.so"

MIRT_RENDER_COUNT_WORK = 1
MIRT_RENDER_TIME_KERNELS = 2
# opt-in estimator changes (none has a counterpart in the reference; all off by default)
MIRT_RENDER_EVEN_LIGHTS = 4
MIRT_RENDER_DROP_NONFINITE = 8


def MIRT_RENDER_STRATA(n):
    """Render flag bits that stratify the pixel jitter over n x n cells (n in 1..255)."""
    if not 0 <= n <= 255:
        raise ValueError("strata per axis must be in 0..255")
    return n << 8

MIRT_KERNEL_RENDER, MIRT_KERNEL_RESOLVE, MIRT_KERNEL_COUNT = 0, 1, 4


class MirtError(RuntimeError):
    pass


# ---- ctypes mirrors of include/mirt.h -------------------------------------------------
class PrimHeader(C.Structure):
    _fields_ = [("material", C.c_uint32), ("chain", C.c_uint32), ("order", C.c_uint32),
                ("type_index", C.c_uint32), ("object_index", C.c_uint32), ("flags", C.c_uint32)]


class Sphere(C.Structure):
    _fields_ = [("h", PrimHeader), ("center", C.c_float * 3), ("radius", C.c_float)]


class MovingSphere(C.Structure):
    _fields_ = [("h", PrimHeader), ("center0", C.c_float * 3), ("center1", C.c_float * 3),
                ("time0", C.c_float), ("time1", C.c_float), ("radius", C.c_float)]


class Rect(C.Structure):
    _fields_ = [("h", PrimHeader), ("axis", C.c_uint32), ("a0", C.c_float), ("a1", C.c_float),
                ("b0", C.c_float), ("b1", C.c_float), ("k", C.c_float)]


class Box(C.Structure):
    _fields_ = [("h", PrimHeader), ("min", C.c_float * 3), ("max", C.c_float * 3)]


class ChainOp(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("p", C.c_float * 3)]


class Chain(C.Structure):
    _fields_ = [("first_op", C.c_uint32), ("n_ops", C.c_uint32)]


class Medium(C.Structure):
    _fields_ = [("boundary_tag", C.c_uint32), ("boundary_index", C.c_uint32), ("neg_inv_density", C.c_float),
                ("material", C.c_uint32), ("chain", C.c_uint32), ("order", C.c_uint32),
                ("type_index", C.c_uint32), ("object_index", C.c_uint32)]


class Material(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("texture", C.c_uint32), ("param", C.c_float)]


class Texture(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("rgb", C.c_float * 3), ("odd", C.c_uint32), ("even", C.c_uint32),
                ("scale", C.c_float), ("index", C.c_uint32)]


class Perlin(C.Structure):
    _fields_ = [("random_vec", (C.c_float * 3) * 256), ("perm_x", C.c_uint32 * 256),
                ("perm_y", C.c_uint32 * 256), ("perm_z", C.c_uint32 * 256)]


class Image(C.Structure):
    _fields_ = [("rgb8", C.c_void_p), ("width", C.c_uint32), ("height", C.c_uint32)]


class Light(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("p", C.c_float * 5)]


class LightGroup(C.Structure):
    _fields_ = [("first_light", C.c_uint32), ("n_lights", C.c_uint32)]


class SceneDesc(C.Structure):
    _fields_ = [("spheres", C.POINTER(Sphere)), ("n_spheres", C.c_uint32),
                ("moving_spheres", C.POINTER(MovingSphere)), ("n_moving_spheres", C.c_uint32),
                ("rects", C.POINTER(Rect)), ("n_rects", C.c_uint32),
                ("boxes", C.POINTER(Box)), ("n_boxes", C.c_uint32),
                ("chains", C.POINTER(Chain)), ("n_chains", C.c_uint32),
                ("chain_ops", C.POINTER(ChainOp)), ("n_chain_ops", C.c_uint32),
                ("media", C.POINTER(Medium)), ("n_media", C.c_uint32),
                ("materials", C.POINTER(Material)), ("n_materials", C.c_uint32),
                ("textures", C.POINTER(Texture)), ("n_textures", C.c_uint32),
                ("perlins", C.POINTER(Perlin)), ("n_perlins", C.c_uint32),
                ("images", C.POINTER(Image)), ("n_images", C.c_uint32),
                ("lights_mode", C.c_uint32),
                ("lights", C.POINTER(Light)), ("n_lights", C.c_uint32),
                ("light_groups", C.POINTER(LightGroup)), ("n_light_groups", C.c_uint32),
                ("time0", C.c_float), ("time1", C.c_float), ("build_flags", C.c_uint32)]


class Camera(C.Structure):
    _fields_ = [("origin", C.c_float * 3), ("lower_left_corner", C.c_float * 3), ("horizontal", C.c_float * 3),
                ("vertical", C.c_float * 3), ("u", C.c_float * 3), ("v", C.c_float * 3), ("w", C.c_float * 3),
                ("lens_radius", C.c_float), ("time0", C.c_float), ("time1", C.c_float)]


class RenderParams(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("samples_per_pixel", C.c_uint32),
                ("first_sample", C.c_uint32), ("max_depth", C.c_uint32), ("background", C.c_float * 3),
                ("seed", C.c_uint64), ("path_pool", C.c_uint32), ("flags", C.c_uint32),
                ("first_row", C.c_uint32), ("n_rows", C.c_uint32)]


class Stats(C.Structure):
    _fields_ = [("render_ms", C.c_double), ("samples", C.c_uint64), ("segments", C.c_uint64),
                ("nodes_visited", C.c_uint64), ("prims_tested", C.c_uint64), ("iterations", C.c_uint32),
                ("path_pool", C.c_uint32), ("kernel_ms", C.c_double * MIRT_KERNEL_COUNT),
                ("kernel_launches", C.c_uint32 * MIRT_KERNEL_COUNT),
                ("phase_cycles", C.c_uint64 * 6),
                ("node_wave_steps", C.c_uint64), ("prim_wave_steps", C.c_uint64),
                ("scene_in_lds", C.c_uint32), ("lds_bytes", C.c_uint32)]

    def as_dict(self):
        return {"render_ms": self.render_ms, "samples": int(self.samples), "segments": int(self.segments),
                "nodes_visited": int(self.nodes_visited), "prims_tested": int(self.prims_tested),
                "iterations": int(self.iterations), "path_pool": int(self.path_pool),
                "kernel_ms": [float(x) for x in self.kernel_ms],
                "kernel_launches": [int(x) for x in self.kernel_launches],
                "phase_cycles": [int(x) for x in self.phase_cycles],
                "node_wave_steps": int(self.node_wave_steps), "prim_wave_steps": int(self.prim_wave_steps),
                "scene_in_lds": bool(self.scene_in_lds), "lds_bytes": int(self.lds_bytes)}


class Ray(C.Structure):
    _fields_ = [("origin", C.c_float * 3), ("direction", C.c_float * 3), ("time", C.c_float)]


MIRT_SYMBOLS = ["mirt_context_create", "mirt_context_create_multi", "mirt_context_device_count", "mirt_context_destroy", "mirt_last_error", "mirt_scene_upload",
                "mirt_scene_destroy", "mirt_render", "mirt_render_accumulate", "mirt_render_accumulate_static",
                "mirt_static_workspace_paths", "mirt_resolve", "mirt_frame_create", "mirt_frame_destroy",
                "mirt_frame_render", "mirt_frame_samples", "mirt_frame_read", "mirt_frame_save", "mirt_frame_load",
                "mirt_trace_primary", "mirt_trace_records", "mirt_trace_device", "mirt_version"]


def build(verbose=False):
    """Compiles libmirt.so (hipcc, gfx950) and librt_examples.so in-tree."""
    out = None if verbose else subprocess.DEVNULL
    subprocess.run(["make", "-C", str(PKG_DIR / "csrc")], check=True, stdout=out)
    subprocess.run(["make", "-C", str(PKG_DIR / "host")], check=True, stdout=out)


_mirt = None
_examples = None


def mirt():
    """The C ABI (include/mirt.h).  Raises if the library has not been built."""
    global _mirt
    if _mirt is not None:
        return _mirt
    if not LIBMIRT.exists():
        raise MirtError(f"{LIBMIRT} is missing: run raytracing_amd.build() (needs hipcc)")
    L = C.CDLL(str(LIBMIRT), mode=C.RTLD_GLOBAL)
    vp = C.c_void_p
    L.mirt_context_create.restype = C.c_int
    L.mirt_context_create.argtypes = [C.c_int, C.POINTER(vp)]
    L.mirt_context_create_multi.restype = C.c_int
    L.mirt_context_create_multi.argtypes = [C.POINTER(C.c_int), C.c_int, C.c_uint32, C.POINTER(vp)]
    L.mirt_context_device_count.restype = C.c_int
    L.mirt_context_device_count.argtypes = [vp]
    L.mirt_context_destroy.argtypes = [vp]
    L.mirt_last_error.restype = C.c_char_p
    L.mirt_last_error.argtypes = [vp]
    L.mirt_scene_upload.restype = C.c_int
    L.mirt_scene_upload.argtypes = [vp, C.POINTER(SceneDesc), C.POINTER(vp)]
    L.mirt_scene_destroy.argtypes = [vp, vp]
    L.mirt_render.restype = C.c_int
    L.mirt_render.argtypes = [vp, vp, C.POINTER(Camera), C.POINTER(RenderParams), vp, vp, C.POINTER(Stats)]
    L.mirt_render_accumulate.restype = C.c_int
    L.mirt_render_accumulate.argtypes = [vp, vp, C.POINTER(Camera), C.POINTER(RenderParams), vp, vp, C.POINTER(Stats)]
    L.mirt_render_accumulate_static.restype = C.c_int
    L.mirt_render_accumulate_static.argtypes = L.mirt_render_accumulate.argtypes
    L.mirt_static_workspace_paths.restype = C.c_int
    L.mirt_resolve.restype = C.c_int
    L.mirt_resolve.argtypes = [vp, vp, C.c_uint32, C.c_uint32, C.c_uint32, vp, vp]
    L.mirt_trace_primary.restype = C.c_int
    L.mirt_trace_primary.argtypes = [vp, vp, vp, C.c_size_t, C.c_uint64, vp, vp, vp]
    L.mirt_frame_create.restype = C.c_int
    L.mirt_frame_create.argtypes = [vp, C.c_uint32, C.c_uint32, C.POINTER(vp)]
    L.mirt_frame_destroy.argtypes = [vp, vp]
    L.mirt_frame_render.restype = C.c_int
    L.mirt_frame_render.argtypes = [vp, vp, C.POINTER(Camera), C.POINTER(RenderParams), vp, C.POINTER(Stats)]
    L.mirt_frame_samples.restype = C.c_uint32
    L.mirt_frame_samples.argtypes = [vp]
    L.mirt_frame_read.restype = C.c_int
    L.mirt_frame_read.argtypes = [vp, vp, vp, vp]
    L.mirt_frame_save.restype = C.c_int
    L.mirt_frame_save.argtypes = [vp, vp, C.c_char_p]
    L.mirt_frame_load.restype = C.c_int
    L.mirt_frame_load.argtypes = [vp, C.c_char_p, C.POINTER(vp)]
    L.mirt_trace_records.restype = C.c_int
    L.mirt_trace_records.argtypes = [vp, vp, vp, C.c_size_t, C.c_uint64, vp]
    L.mirt_trace_device.restype = C.c_int
    L.mirt_trace_device.argtypes = [vp, vp, vp, C.c_size_t, C.c_uint64, vp, vp, vp, vp]
    L.mirt_version.restype = C.c_int
    _mirt = L
    return L


def examples():
    """The host mirror's example scenes (librt_examples.so)."""
    global _examples
    if _examples is not None:
        return _examples
    mirt()
    if not LIBEXAMPLES.exists():
        raise MirtError(f"{LIBEXAMPLES} is missing: run raytracing_amd.build()")
    L = C.CDLL(str(LIBEXAMPLES))
    vp = C.c_void_p
    L.rt_example_last_error.restype = C.c_char_p
    L.rt_example_open.restype = vp
    L.rt_example_open.argtypes = [C.c_char_p, C.c_uint64, C.c_char_p, C.c_int]
    L.rt_example_close.argtypes = [vp]
    L.rt_example_defaults.argtypes = [vp, C.POINTER(C.c_uint32), C.POINTER(C.c_float)]
    L.rt_example_primitive_count.restype = C.c_uint64
    L.rt_example_primitive_count.argtypes = [vp]
    L.rt_example_object_count.restype = C.c_uint64
    L.rt_example_object_count.argtypes = [vp]
    L.rt_example_context.restype = vp
    L.rt_example_context.argtypes = [vp]
    L.rt_example_scene.restype = vp
    L.rt_example_scene.argtypes = [vp]
    L.rt_example_camera.argtypes = [vp, C.c_float, C.POINTER(Camera), C.POINTER(C.c_float)]
    L.rt_example_render.restype = C.c_int
    L.rt_example_render.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_float,
                                    C.c_uint64, C.c_uint32, C.c_uint32, vp, vp, C.POINTER(Stats)]
    L.rt_example_render_as_reference.restype = C.c_int
    L.rt_example_render_as_reference.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_float,
                                                 C.c_uint64, C.c_uint32, vp, C.POINTER(Stats)]
    L.rt_reference_strips_place.restype = C.c_int
    L.rt_reference_strips_place.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_uint32]
    L.rt_example_render_accumulate.restype = C.c_int
    L.rt_example_render_accumulate.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                               C.c_float, C.c_uint64, C.c_uint32, C.c_uint32, vp, vp,
                                               C.POINTER(Stats)]
    L.rt_example_resolve.restype = C.c_int
    L.rt_example_resolve.argtypes = [vp, vp, C.c_uint32, C.c_uint32, C.c_uint32, vp, vp]
    L.rt_lowered_open.restype = vp
    L.rt_lowered_open.argtypes = [C.c_char_p, C.c_uint64, C.c_char_p]
    L.rt_lowered_close.argtypes = [vp]
    L.rt_lowered_upload.restype = C.c_int
    L.rt_lowered_upload.argtypes = [vp, vp, C.POINTER(vp)]
    L.rt_example_trace.restype = C.c_int
    L.rt_example_trace.argtypes = [vp, vp, C.c_size_t, C.c_uint64, vp, vp, vp]
    _examples = L
    return L


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class ExampleScene:
    """One of the reference's example scenes, lowered and resident on a GPU."""

    def __init__(self, name, scene_seed=1, device=0, asset_dir=None):
        self.L = examples()
        self.name = name
        self.h = self.L.rt_example_open(name.encode(), scene_seed, str(asset_dir or ASSET_DIR).encode(), device)
        if not self.h:
            raise MirtError(self.L.rt_example_last_error().decode())
        d = (C.c_uint32 * 4)()
        a = C.c_float()
        self.L.rt_example_defaults(self.h, d, C.byref(a))
        self.width, self.height, self.spp, self.depth = (int(x) for x in d)
        self.aspect = float(a.value)

    def close(self):
        if getattr(self, "h", None):
            self.L.rt_example_close(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def _check(self, rc):
        if rc != 0:
            raise MirtError(f"status {rc}: {self.L.rt_example_last_error().decode()}")

    def primitive_count(self):
        return int(self.L.rt_example_primitive_count(self.h))

    def object_count(self):
        return int(self.L.rt_example_object_count(self.h))

    def camera(self, aspect=0.0):
        cam = Camera()
        bg = (C.c_float * 3)()
        self.L.rt_example_camera(self.h, aspect, C.byref(cam), bg)
        return cam, [float(x) for x in bg]

    def render(self, width, height, spp, depth=None, aspect=0.0, seed=7, first_sample=0, path_pool=0, flags=0):
        """mirt_render: returns (rgb24 HxWx3 u8, sums HxWx3 f32, stats dict)."""
        rgb = np.zeros((height, width, 3), np.uint8)
        sums = np.zeros((height, width, 3), np.float32)
        st = Stats()
        self._check(self.L.rt_example_render(self.h, width, height, spp, first_sample,
                                             self.depth if depth is None else depth, aspect, seed, path_pool,
                                             flags, _ptr(rgb), _ptr(sums), C.byref(st)))
        return rgb, sums, st.as_dict()

    def render_as_reference(self, width, height, spp, threads, depth=None, aspect=0.0, seed=7):
        """The frame as the reference leaves it on `threads` threads (renderer.rs:55-87): height % threads rows
        never rendered (black, at the end), the image that many rows higher.  Returns (rgb24 HxWx3 u8, stats)."""
        rgb = np.zeros((height, width, 3), np.uint8)
        st = Stats()
        self._check(self.L.rt_example_render_as_reference(self.h, width, height, spp,
                                                          self.depth if depth is None else depth, aspect, seed,
                                                          threads, _ptr(rgb), C.byref(st)))
        return rgb, st.as_dict()

    def render_accumulate(self, sum_ptr, width, height, spp, depth=None, aspect=0.0, seed=7, first_sample=0,
                          path_pool=0, flags=0, stream=None):
        """mirt_render_accumulate into a device buffer (e.g. torch tensor .data_ptr())."""
        st = Stats()
        self._check(self.L.rt_example_render_accumulate(self.h, width, height, spp, first_sample,
                                                        self.depth if depth is None else depth, aspect, seed,
                                                        path_pool, flags, C.c_void_p(sum_ptr),
                                                        C.c_void_p(stream or 0), C.byref(st)))
        return st.as_dict()

    def resolve(self, sum_ptr, width, height, spp, rgb_ptr, stream=None):
        self._check(self.L.rt_example_resolve(self.h, C.c_void_p(sum_ptr), width, height, spp, C.c_void_p(rgb_ptr),
                                              C.c_void_p(stream or 0)))

    def trace(self, rays, seed=7):
        """mirt_trace_primary: rays n x 7 f32 -> (type_index, object_index, t)."""
        rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 7)
        n = rays.shape[0]
        ti = np.zeros(n, np.uint32)
        oi = np.zeros(n, np.uint32)
        t = np.zeros(n, np.float32)
        self._check(self.L.rt_example_trace(self.h, _ptr(rays), n, seed, _ptr(ti), _ptr(oi), _ptr(t)))
        return ti, oi, t


class SharedScene:
    """An example scene uploaded through one context over several GPUs (mirt_context_create_multi): mirt_render then
    shares the frame out between them inside the library -- row bands x sample ranges, one host thread per device,
    the other devices' rows copied to the first one and added there (the counterpart of the reference's strip
    threads, renderer.rs:55-87)."""

    def __init__(self, name, devices, row_bands=0, scene_seed=1, asset_dir=None):
        self.L = mirt()
        self.X = examples()
        self.ctx = C.c_void_p()
        self.scene = C.c_void_p()
        self.lowered = self.X.rt_lowered_open(name.encode(), scene_seed, str(asset_dir or ASSET_DIR).encode())
        if not self.lowered:
            raise MirtError(self.X.rt_example_last_error().decode())
        ids = (C.c_int * len(devices))(*devices)
        rc = self.L.mirt_context_create_multi(ids, len(devices), row_bands, C.byref(self.ctx))
        if rc != 0:
            raise MirtError(f"status {rc}: {self.L.mirt_last_error(None).decode()}")
        rc = self.X.rt_lowered_upload(self.lowered, self.ctx, C.byref(self.scene))
        if rc != 0:
            raise MirtError(f"status {rc}: {self.L.mirt_last_error(self.ctx).decode()}")

    def render(self, cam, background, width, height, spp, depth, seed=7, first_sample=0, first_row=0, n_rows=0,
               flags=0, path_pool=0, want_sums=True):
        """mirt_render into host buffers: (rgb24 HxWx3 u8, sums HxWx3 f32 or None, stats dict)."""
        p = RenderParams(width, height, spp, first_sample, depth, (C.c_float * 3)(*background), seed, path_pool, flags,
                         first_row, n_rows)
        rgb = np.zeros((height, width, 3), np.uint8)
        sums = np.zeros((height, width, 3), np.float32) if want_sums else None
        st = Stats()
        rc = self.L.mirt_render(self.ctx, self.scene, C.byref(cam), C.byref(p), rgb.ctypes.data,
                                sums.ctypes.data if want_sums else None, C.byref(st))
        if rc != 0:
            raise MirtError(f"status {rc}: {self.L.mirt_last_error(self.ctx).decode()}")
        return rgb, sums, st.as_dict()

    def close(self):
        if self.scene:
            self.L.mirt_scene_destroy(self.ctx, self.scene)
            self.scene = C.c_void_p()
        if self.ctx:
            self.L.mirt_context_destroy(self.ctx)
            self.ctx = C.c_void_p()
        if self.lowered:
            self.X.rt_lowered_close(self.lowered)
            self.lowered = None

    def __del__(self):
        self.close()


RECORD_DTYPE = np.dtype([("point", np.float32, 3), ("t", np.float32), ("normal", np.float32, 3),
                         ("front_face", np.uint32), ("u", np.float32), ("v", np.float32), ("type_index", np.uint32),
                         ("object_index", np.uint32), ("color", np.float32, 3), ("material_kind", np.uint32)])


def trace_records(example, rays, seed=7):
    """mirt_trace_records on an ExampleScene: rays n x 7 f32 -> structured array in the layout of mirt_hit_record."""
    rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 7)
    out = np.zeros(rays.shape[0], RECORD_DTYPE)
    L = mirt()
    ctx, scene = example.L.rt_example_context(example.h), example.L.rt_example_scene(example.h)
    rc = L.mirt_trace_records(ctx, scene, _ptr(rays), rays.shape[0], seed, _ptr(out))
    if rc != 0:
        raise MirtError(f"status {rc}: {L.mirt_last_error(ctx).decode()}")
    return out


def shard_samples(spp, world_size, rank):
    """Sample-range sharding: rank r renders samples [first, first + count) of every pixel."""
    base, extra = divmod(spp, world_size)
    return rank * base + min(rank, extra), base + (1 if rank < extra else 0)


def camera_rays(cam, width, height, time=0.5):
    """Pixel-centre rays of a mirt Camera, lens offset zero; ray i = y*W + x, y up (camera.rs:61-62 in f32)."""
    f = np.float32
    origin = np.array(list(cam.origin), f)
    llc = np.array(list(cam.lower_left_corner), f)
    hor = np.array(list(cam.horizontal), f)
    ver = np.array(list(cam.vertical), f)
    xs = (np.arange(width, dtype=f) + f(0.5)) / f(width - 1)
    ys = (np.arange(height, dtype=f) + f(0.5)) / f(height - 1)
    u = np.tile(xs, height)[:, None]
    v = np.repeat(ys, width)[:, None]
    zero = np.zeros(3, f)
    d = (((llc[None, :] + u * hor[None, :]) + v * ver[None, :]) - origin[None, :]) - zero[None, :]
    rays = np.zeros((width * height, 7), f)
    rays[:, 0:3] = origin + zero
    rays[:, 3:6] = d.astype(f)
    rays[:, 6] = time
    return rays
