"""GPU tests that drive include/mirt.h directly with hand-built scene descriptions (edge cases and errors)."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as O
import raytracing_amd as R

pytestmark = pytest.mark.gpu
F = np.float32


class Ctx:
    def __init__(self):
        self.L = R.mirt()
        self.ctx = C.c_void_p()
        rc = self.L.mirt_context_create(0, C.byref(self.ctx))
        assert rc == 0, self.L.mirt_last_error(None)
        self.scenes = []

    def close(self):
        for s in self.scenes:
            self.L.mirt_scene_destroy(self.ctx, s)
        self.L.mirt_context_destroy(self.ctx)

    def error(self):
        return self.L.mirt_last_error(self.ctx).decode()

    def upload(self, spheres=(), rects=(), boxes=(), moving=(), materials=None, textures=None, chains=None,
               chain_ops=(), media=(), time=(0.0, 1.0), expect=0, lights=(), light_groups=(), lights_mode=0):
        keep = []

        def arr(T, items):
            a = (T * max(1, len(items)))(*items)
            keep.append(a)
            return a
        d = R.SceneDesc()
        d.spheres, d.n_spheres = arr(R.Sphere, spheres), len(spheres)
        d.moving_spheres, d.n_moving_spheres = arr(R.MovingSphere, moving), len(moving)
        d.rects, d.n_rects = arr(R.Rect, rects), len(rects)
        d.boxes, d.n_boxes = arr(R.Box, boxes), len(boxes)
        chains = chains if chains is not None else [R.Chain(0, 0)]
        d.chains, d.n_chains = arr(R.Chain, chains), len(chains)
        d.chain_ops, d.n_chain_ops = arr(R.ChainOp, chain_ops), len(chain_ops)
        materials = materials if materials is not None else [R.Material(0, 0, 0.0)]
        textures = textures if textures is not None else [R.Texture(0, (C.c_float * 3)(0.5, 0.5, 0.5), 0, 0, 0.0, 0)]
        d.materials, d.n_materials = arr(R.Material, materials), len(materials)
        d.textures, d.n_textures = arr(R.Texture, textures), len(textures)
        d.media, d.n_media = arr(R.Medium, media), len(media)
        d.lights, d.n_lights = arr(R.Light, lights), len(lights)
        d.light_groups, d.n_light_groups = arr(R.LightGroup, light_groups), len(light_groups)
        d.lights_mode = lights_mode
        d.time0, d.time1 = time
        scene = C.c_void_p()
        rc = self.L.mirt_scene_upload(self.ctx, C.byref(d), C.byref(scene))
        assert rc == expect, (rc, self.error())
        if rc == 0:
            self.scenes.append(scene)
        return scene

    def trace(self, scene, rays, expect=0):
        rays = np.ascontiguousarray(rays, F).reshape(-1, 7)
        n = len(rays)
        ti, oi, t = np.zeros(n, np.uint32), np.zeros(n, np.uint32), np.zeros(n, F)
        rc = self.L.mirt_trace_primary(self.ctx, scene, rays.ctypes.data, n, 0, ti.ctypes.data, oi.ctypes.data,
                                       t.ctypes.data)
        assert rc == expect, (rc, self.error())
        return ti, oi, t

    def render(self, scene, w, h, spp, depth=5, bg=(0.25, 0.5, 1.0), expect=0, cam=None, flags=0):
        cam = cam or simple_camera()
        p = R.RenderParams(w, h, spp, 0, depth, (C.c_float * 3)(*bg), 9, 0, flags)
        rgb = np.zeros((h, w, 3), np.uint8)
        sums = np.zeros((h, w, 3), F)
        st = R.Stats()
        rc = self.L.mirt_render(self.ctx, scene, C.byref(cam), C.byref(p), rgb.ctypes.data, sums.ctypes.data,
                                C.byref(st))
        assert rc == expect, (rc, self.error())
        return rgb, sums, st


def simple_camera(time0=0.0, time1=1.0):
    # looks down -z from the origin: lower-left corner (-1,-1,-1), 2x2 viewport
    c = R.Camera()
    c.origin[:] = [0, 0, 0]
    c.lower_left_corner[:] = [-1, -1, -1]
    c.horizontal[:] = [2, 0, 0]
    c.vertical[:] = [0, 2, 0]
    c.u[:] = [1, 0, 0]
    c.v[:] = [0, 1, 0]
    c.w[:] = [0, 0, 1]
    c.lens_radius, c.time0, c.time1 = 0.0, time0, time1
    return c


def hdr(order, ti=0, oi=0, material=0, chain=0, flags=0):
    return R.PrimHeader(material, chain, order, ti, oi, flags)


def sphere(c, r, order, ti=0, oi=0, **kw):
    return R.Sphere(hdr(order, ti, oi, **kw), (C.c_float * 3)(*c), r)


@pytest.fixture()
def ctx():
    c = Ctx()
    yield c
    c.close()


def test_empty_scene_renders_background_and_traces_misses(ctx):
    scene = ctx.upload()
    ti, oi, t = ctx.trace(scene, [[0, 0, 0, 0, 0, -1, 0.5]] * 3)
    assert np.all(ti == 0xFFFFFFFF) and np.all(oi == 0xFFFFFFFF) and np.all(np.isinf(t))
    rgb, sums, st = ctx.render(scene, 16, 8, 3)
    assert np.allclose(sums, np.array([0.25, 0.5, 1.0], F) * 3)
    assert np.array_equal(rgb, O.encode(sums, 3))


def test_single_primitive_known_answer(ctx):
    """objects.rs:42-59 by hand: unit sphere, ray from (0,0,-5) along +z hits at t = 4."""
    scene = ctx.upload(spheres=[sphere((0, 0, 0), 1.0, 0, ti=3, oi=7)])
    ti, oi, t = ctx.trace(scene, [[0, 0, -5, 0, 0, 1, 0], [0, 2, -5, 0, 0, 1, 0], [0, 0, -5, 0, 0, 2, 0],
                                  [0, 0, 0, 1, 0, 0, 0]])
    assert list(ti) == [3, 0xFFFFFFFF, 3, 3] and list(oi) == [7, 0xFFFFFFFF, 7, 7]
    assert t[0] == 4.0 and np.isinf(t[1]) and t[2] == 2.0 and t[3] == 1.0


def test_equal_t_goes_to_the_higher_order(ctx):
    """Two identical spheres: world.rs:75-94 lets the later one replace an equal t; bvh.rs:122-126 the right leaf."""
    for first_wins in (False, True):
        a = sphere((0, 0, -3), 1.0, 5 if first_wins else 1, ti=0, oi=0)
        b = sphere((0, 0, -3), 1.0, 2, ti=1, oi=0)
        scene = ctx.upload(spheres=[a, b])
        ti, _, t = ctx.trace(scene, [[0, 0, 0, 0, 0, -1, 0]])
        assert t[0] == 2.0 and ti[0] == (0 if first_wins else 1)


def test_many_coincident_primitives_do_not_overflow_the_traversal_stack(ctx):
    spheres = [sphere((0, 0, -3), 1.0, i, ti=0, oi=i) for i in range(300)]
    scene = ctx.upload(spheres=spheres)
    _, oi, t = ctx.trace(scene, [[0, 0, 0, 0, 0, -1, 0]])
    assert t[0] == 2.0 and oi[0] == 299


def test_rect_and_box_known_answers(ctx):
    rect = R.Rect(hdr(0, 0, 0), 1, 0.0, 2.0, 0.0, 4.0, 1.0)          # XZ rect at y = 1
    box = R.Box(hdr(1, 1, 0), (C.c_float * 3)(5, 0, 0), (C.c_float * 3)(6, 2, 3))
    scene = ctx.upload(rects=[rect], boxes=[box])
    ti, _, t = ctx.trace(scene, [[2, 0, 1, 0, 1, 0, 0],        # through the rect's edge x == x1: hit
                                 [2.0001, 0, 1, 0, 1, 0, 0],   # just outside
                                 [5.5, 1, -1, 0, 0, 1, 0],     # box front face
                                 [5.5, 1, 1, 1, 0, 0, 0]])     # from inside the box
    assert list(ti) == [0, 0xFFFFFFFF, 1, 1]
    assert t[0] == 1.0 and t[2] == 1.0 and t[3] == 0.5


def test_translate_and_rotate_wrappers(ctx):
    """A sphere at the origin wrapped in Translate((0,0,-4)) is hit like a sphere at (0,0,-4) (transforms.rs:15-24);
    Rotate by 90 degrees about y maps the object-space point (0,0,3) to world (3,0,0) (transforms.rs:97-101)."""
    ops = [R.ChainOp(0, (C.c_float * 3)(0, 0, -4)), R.ChainOp(1, (C.c_float * 3)(1.0, 0.0, 0.0))]
    chains = [R.Chain(0, 0), R.Chain(0, 1), R.Chain(1, 1)]
    scene = ctx.upload(spheres=[sphere((0, 0, 0), 1.0, 0, ti=0, chain=1), sphere((0, 0, 3), 0.5, 1, ti=1, chain=2)],
                       chains=chains, chain_ops=ops)
    ti, _, t = ctx.trace(scene, [[0, 0, 0, 0, 0, -1, 0], [0, 0, 0, 1, 0, 0, 0], [0, 0, 0, 0, 0, 1, 0]])
    assert list(ti) == [0, 1, 0xFFFFFFFF]
    assert t[0] == 3.0 and abs(t[1] - 2.5) < 1e-6


def test_moving_sphere_time_range_is_enforced(ctx):
    m = R.MovingSphere(hdr(0), (C.c_float * 3)(0, 0, -3), (C.c_float * 3)(0, 2, -3), 0.0, 1.0, 0.5)
    scene = ctx.upload(moving=[m], time=(0.0, 1.0))
    ti, _, t = ctx.trace(scene, [[0, 0, 0, 0, 0, -1, 0.0], [0, 0, 0, 0, 0, -1, 1.0], [0, 2, 0, 0, 0, -1, 1.0]])
    assert list(ti) == [0, 0xFFFFFFFF, 0] and t[0] == 2.5 and t[2] == 2.5
    ctx.trace(scene, [[0, 0, 0, 0, 0, -1, 1.5]], expect=1)
    assert "time" in ctx.error()
    ctx.render(scene, 8, 8, 1, cam=simple_camera(0.0, 2.0), expect=1)


def test_invalid_arguments_are_rejected_with_a_message(ctx):
    L = ctx.L
    assert L.mirt_scene_upload(ctx.ctx, None, None) == 1
    ctx.upload(spheres=[sphere((0, 0, 0), 1.0, 0, material=3)], expect=1)
    assert "material" in ctx.error()
    ctx.upload(chains=[R.Chain(0, 1)], expect=1)                      # chains[0] must be empty
    ctx.upload(chains=[R.Chain(0, 0), R.Chain(0, 5)], chain_ops=[R.ChainOp(0, (C.c_float * 3)(0, 0, 0))] * 5, expect=3)
    ctx.upload(spheres=[sphere((np.inf, 0, 0), 1.0, 0)], expect=1)
    scene = ctx.upload(spheres=[sphere((0, 0, -3), 1.0, 0)])
    ctx.render(scene, 1, 8, 1, expect=1)
    ctx.render(scene, 8, 8, 0, expect=1)
    assert "samples_per_pixel" in ctx.error()
    bad = C.c_void_p()
    assert L.mirt_context_create(99, C.byref(bad)) == 1 and not bad.value
    assert L.mirt_last_error(None)


def test_render_of_one_emissive_sphere(ctx):
    """A diffuse light of radiance 2 filling the view: every pixel sums to spp * 2 (material.rs:151-158)."""
    mats = [R.Material(3, 0, 0.0)]
    tex = [R.Texture(0, (C.c_float * 3)(2.0, 2.0, 2.0), 0, 0, 0.0, 0)]
    scene = ctx.upload(spheres=[sphere((0, 0, -3), 2.9, 0)], materials=mats, textures=tex)
    rgb, sums, st = ctx.render(scene, 8, 8, 5)
    assert np.allclose(sums, 10.0)
    assert np.all(rgb == 255)


def test_constant_medium_inside_a_box_boundary(ctx):
    """ConstantMedium over a Box3d boundary, the smoke box of a Cornell scene (transforms.rs:134-180).  With a
    density so high that -1/density is -0 the free flight is 0 whatever the draw, so the ray scatters exactly
    where it enters the boundary; the boundary itself is not a surface.  A Translate around the medium moves it."""
    dense = -0.0
    box = R.Box(hdr(0, 0, 0, flags=1), (C.c_float * 3)(-1, -1, -6), (C.c_float * 3)(1, 1, -4))     # boundary only
    far = sphere((0, 0, -20), 2.0, order=1, ti=1, oi=0)
    mats = [R.Material(0, 0, 0.0), R.Material(4, 0, 0.0)]                                             # lambertian, isotropic
    smoke = R.Medium(3, 0, dense, 1, 0, 2, 7, 3)                                                        # box 0, WorldIndex (7, 3)
    scene = ctx.upload(spheres=[far], boxes=[box], materials=mats, media=[smoke])
    ti, oi, t = ctx.trace(scene, [[0, 0, 0, 0, 0, -1, 0],        # enters the box at z = -4
                                  [0, 0, -5, 0, 0, -1, 0],       # starts inside: scatters at once (t_min)
                                  [3, 0, 0, 0, 0, -1, 0],        # passes beside the box: nothing on its way
                                  [0, 0, -30, 0, 0, 1, 0]])      # from behind: the sphere is hit first, t = 8
    assert list(ti) == [7, 7, 0xFFFFFFFF, 1] and list(oi[:2]) == [3, 3]
    assert t[0] == 4.0 and t[1] == F(0.001) and t[3] == 8.0
    # the same smoke box wrapped in Translate((0, 0, -2)): the boundary's chain and the medium's chain
    ops = [R.ChainOp(0, (C.c_float * 3)(0, 0, -2))]
    chains = [R.Chain(0, 0), R.Chain(0, 1)]
    box2 = R.Box(hdr(0, 0, 0, chain=1, flags=1), (C.c_float * 3)(-1, -1, -6), (C.c_float * 3)(1, 1, -4))
    smoke2 = R.Medium(3, 0, dense, 1, 1, 2, 7, 3)
    scene2 = ctx.upload(spheres=[far], boxes=[box2], materials=mats, media=[smoke2], chains=chains, chain_ops=ops)
    ti, _, t = ctx.trace(scene2, [[0, 0, 0, 0, 0, -1, 0]])
    assert ti[0] == 7 and t[0] == 6.0
    # rendered: every camera ray that meets the smoke ends there (isotropic: no scatter, no emission) -> black;
    # the others see the background or the grey sphere
    rgb, sums, _ = ctx.render(scene, 32, 32, 4, bg=(1.0, 1.0, 1.0))
    assert np.all(sums[14:18, 14:18] == 0.0)
    assert np.all(sums[0, :] > 0.0)


def test_long_light_lists_sample_the_same_light(ctx):
    """A lit floor under one downward-facing rect light, with the light list given three ways: the bare object, a
    World of 12 copies of it in one type group (longer than the list the kernel keeps on chip), and that World
    sampled evenly.  Copies of one light change neither World::random nor the mean of pdf_value
    (world.rs:100-117), so the three renders estimate the same image."""
    mats = [R.Material(0, 0, 0.0), R.Material(3, 1, 0.0)]                                  # grey lambertian, light
    tex = [R.Texture(0, (C.c_float * 3)(0.5, 0.5, 0.5), 0, 0, 0.0, 0), R.Texture(0, (C.c_float * 3)(8.0, 8.0, 8.0), 0, 0, 0.0, 0)]
    ops = [R.ChainOp(2, (C.c_float * 3)(0, 0, 0))]                                         # FlipFace: emits downwards
    chains = [R.Chain(0, 0), R.Chain(0, 1)]
    floor = R.Rect(hdr(0, 0, 0), 1, -4.0, 4.0, -8.0, 0.0, -1.0)                            # XZ rect at y = -1
    lamp = R.Rect(hdr(1, 1, 0, material=1, chain=1), 1, -1.0, 1.0, -5.0, -3.0, 1.5)        # XZ rect at y = 1.5
    light = R.Light(0, (C.c_float * 5)(-1.0, 1.0, -5.0, -3.0, 1.5))
    kw = dict(rects=[floor, lamp], materials=mats, textures=tex, chains=chains, chain_ops=ops)
    one = ctx.upload(lights=[light], lights_mode=1, **kw)
    many = ctx.upload(lights=[light] * 12, light_groups=[R.LightGroup(0, 12)], lights_mode=2, **kw)
    cam = simple_camera()
    cam.origin[:] = [0, 0.5, 0]
    cam.lower_left_corner[:] = [-1, -1.5, -1]                                              # looks a little downwards
    images = [ctx.render(s, 48, 48, 64, depth=3, bg=(0, 0, 0), cam=cam, flags=f)[1]
              for s, f in ((one, 0), (many, 0), (many, R.MIRT_RENDER_EVEN_LIGHTS))]
    lit = images[0][30:, :, :]                                                             # bottom rows: the floor
    assert np.isfinite(lit).all() and lit.mean() > 0.5
    for other in images[1:]:
        assert np.isfinite(other).all()
        assert not np.array_equal(other, images[0])
        assert abs(other[30:].mean() - lit.mean()) < 0.03 * lit.mean()


@pytest.mark.parametrize("kind,param", [(0, 0.0), (1, 0.0), (2, 1.5)])
def test_a_convex_body_under_a_uniform_sky_has_closed_form_radiance(ctx, kind, param):
    """renderer.rs:140-190 by hand.  One sphere under a constant background B: a scattered ray cannot meet a convex
    body again, so a camera path is worth B when it misses and albedo * B when it hits -- for a Lambertian surface
    scattering_pdf / pdf is exactly 1 (material.rs:50-57 against pdf.rs:25-32), for a mirror the attenuation is the
    albedo (material.rs:75-91) -- and a glass ball (attenuation 1, material.rs:111-137) cannot be seen at all.
    Every pixel therefore sums to k * albedo * B + (spp - k) * B with a whole number k of its samples on the ball."""
    albedo = np.array([0.5, 0.25, 0.75], F)
    bg = np.array([0.25, 0.5, 1.0], F)
    mats = [R.Material(kind, 0, param)]
    tex = [R.Texture(0, (C.c_float * 3)(*albedo), 0, 0, 0.0, 0)]
    scene = ctx.upload(spheres=[sphere((0, 0, -3), 0.75, 0)], materials=mats, textures=tex)
    spp = 4
    _, sums, _ = ctx.render(scene, 32, 32, spp, depth=50, bg=tuple(bg))
    if kind == 2:
        assert np.allclose(sums, spp * bg, rtol=1e-6, atol=0)
        return
    k = (spp * bg[0] - sums[..., 0]) / (bg[0] * (1.0 - albedo[0]))       # samples on the ball, from the red channel
    assert np.all(np.abs(k - np.round(k)) < 1e-4)
    k = np.round(k)[..., None]
    assert k.min() == 0 and k.max() == spp
    assert np.allclose(sums, k * albedo * bg + (spp - k) * bg, rtol=2e-6, atol=0)
    assert np.all(k[14:18, 14:18] == spp) and np.all(k[0] == 0)


@pytest.mark.parametrize("with_sphere", [False, True])
def test_light_sampling_keeps_a_furnace_in_balance(ctx, with_sphere):
    """A floor under a sky of radiance B and a lamp that emits the same B downwards: whatever direction the mixture
    of light and cosine sampling picks (pdf.rs:81-91, objects.rs:292-309), it brings back B, so the floor's expected
    radiance is albedo * B (renderer.rs:166-183) -- a wrong density in either half of the mixture would tilt it.
    with_sphere: the light list is a World of the lamp and a glowing ball, one object of each type as in the
    Cornell box (world.rs:100-117, objects.rs:82-100)."""
    b, albedo, spp = 0.5, 0.5, 256
    mats = [R.Material(0, 0, 0.0), R.Material(3, 1, 0.0)]                                  # lambertian floor, light
    tex = [R.Texture(0, (C.c_float * 3)(albedo, albedo, albedo), 0, 0, 0.0, 0), R.Texture(0, (C.c_float * 3)(b, b, b), 0, 0, 0.0, 0)]
    ops = [R.ChainOp(2, (C.c_float * 3)(0, 0, 0))]                                         # FlipFace: emits downwards
    chains = [R.Chain(0, 0), R.Chain(0, 1)]
    floor = R.Rect(hdr(0, 0, 0), 1, -40.0, 40.0, -80.0, 0.0, -1.0)                         # XZ rect at y = -1
    lamp = R.Rect(hdr(1, 1, 0, material=1, chain=1), 1, -1.0, 1.0, -5.0, -3.0, 1.5)        # XZ rect at y = 1.5
    lights = [R.Light(0, (C.c_float * 5)(-1.0, 1.0, -5.0, -3.0, 1.5))]
    kw = dict(lights=lights, lights_mode=1)
    if with_sphere:
        kw = dict(spheres=[sphere((2.0, 0.0, -4.0), 0.5, 2, ti=2, oi=0, material=1)],
                  lights=lights + [R.Light(1, (C.c_float * 5)(2.0, 0.0, -4.0, 0.5, 0.0))],
                  light_groups=[R.LightGroup(0, 1), R.LightGroup(1, 1)], lights_mode=2)
    scene = ctx.upload(rects=[floor, lamp], materials=mats, textures=tex, chains=chains, chain_ops=ops, **kw)
    cam = simple_camera()
    cam.origin[:] = [0, 0.5, 0]
    cam.lower_left_corner[:] = [-1, -1.5, -1]                                              # looks down at the floor
    _, sums, _ = ctx.render(scene, 48, 48, spp, depth=3, bg=(b, b, b), cam=cam)
    assert np.isfinite(sums).all()
    floor_rows = sums[36:] / spp                                                           # the floor in front of the camera
    assert abs(floor_rows.mean() - albedo * b) < 0.01 * albedo * b, floor_rows.mean()
    per_pixel = floor_rows.mean(axis=2)
    assert np.all(np.abs(per_pixel - albedo * b) < 0.25 * albedo * b)                      # no pixel far off either


def test_an_absorbing_medium_transmits_two_to_the_minus_density_times_chord(ctx):
    """transforms.rs:152-158: the free flight is -1/density * log2(draw) -- base 2 -- so a ray crosses a chord of length
    l of the medium with probability 2^(-density * l), not e^(-density * l).  An Isotropic phase function absorbs
    (material.rs:172-180), so under a white sky the pixel is that probability."""
    rho, radius, spp = 0.5, 1.0, 1024
    ball = sphere((0, 0, -3), radius, 0, flags=1)                                          # boundary only
    mats = [R.Material(0, 0, 0.0), R.Material(4, 0, 0.0)]                                  # (unused), isotropic
    smoke = R.Medium(0, 0, -1.0 / rho, 1, 0, 0, 5, 0)
    scene = ctx.upload(spheres=[ball], materials=mats, media=[smoke])
    n = 32
    _, sums, _ = ctx.render(scene, n, n, spp, depth=5, bg=(1.0, 1.0, 1.0))
    got = sums[..., 0] / spp
    centre = -1.0 + 2.0 * (np.arange(n) + 0.5) / (n - 1)                                   # renderer.rs:216-217
    dx, dy = np.meshgrid(centre, centre[::-1])                                             # row 0 is the top
    d = np.stack([dx, dy, -np.ones_like(dx)], axis=-1)
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    c = np.array([0.0, 0.0, -3.0])
    m2 = np.sum(np.cross(np.broadcast_to(c, d.shape), d) ** 2, axis=-1)                    # squared distance of the line from the centre
    chord = 2.0 * np.sqrt(np.maximum(radius ** 2 - m2, 0.0))
    want = 2.0 ** (-rho * chord)
    assert abs(got[12:20, 12:20].mean() - want[12:20, 12:20].mean()) < 0.01
    assert abs(want[16, 16] - 0.5) < 0.01 and abs(got[14:18, 14:18].mean() - 0.5) < 0.02  # through the centre: 2^-1, not e^-1 = 0.37
    assert np.all(got[0] == 1.0)                                                           # beside the ball: the sky
