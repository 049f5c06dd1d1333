"""Record-level parity: every field of the hit record the device hands to shading (mirt_trace_records, through the
render kernel's own walk and hit_record code) against the oracle's HitRecord for the same rays.

Bars: ids, t, point, normal and front_face bit for bit on surfaces (f32 +,-,*,/,sqrt in the reference's order);
(u, v) of spheres go through acos / atan2 and noise / checker colours through sin, where the device's libm and
glibc may differ in the last place: |du|, |dv| <= 1e-6, colours <= 2e-5 except where such a difference flips an image
texel or a checker cell (counted, at most 0.2 % of the hits on those textures).  Hits inside a constant medium
inherit the 2e-5 relative tolerance of log2 in t and point."""
import numpy as np
import pytest

import oracle_lib as O
import raytracing_amd as R
from test_gpu_parity import random_rays, scenes  # noqa: F401  (fixture)

pytestmark = pytest.mark.gpu
F = np.float32


def bits(a):
    return np.ascontiguousarray(a, F).view(np.uint32)


def compare_records(g, o, rays, medium_type=None, texel_types=()):
    got = R.trace_records(g, rays)
    ref = O.hit_records_full(o, rays)
    same_id = (got["type_index"] == ref["type_index"]) & (got["object_index"] == ref["object_index"])
    medium = np.zeros(len(rays), bool)
    if medium_type is not None:
        medium = (got["type_index"] == medium_type) | (ref["type_index"] == medium_type)
    assert np.all(same_id | medium), "ids differ on a surface"
    hit = ref["last"] == 1
    surf = hit & same_id & ~medium
    assert np.array_equal(np.isinf(got["t"][same_id]), ~hit[same_id])
    for field in ("t", "point", "normal"):
        assert np.array_equal(bits(got[field][surf]), bits(ref[field][surf])), f"{field} differs bitwise on surfaces"
    assert np.array_equal(got["front_face"][surf], ref["front_face"][surf])
    assert np.abs(got["u"][surf] - ref["u"][surf]).max(initial=0) <= 1e-6
    assert np.abs(got["v"][surf] - ref["v"][surf]).max(initial=0) <= 1e-6
    dcol = np.abs(got["color"][surf] - ref["color"][surf]).max(axis=1)
    loose = np.isin(got["type_index"][surf], texel_types)
    assert dcol[~loose].max(initial=0) <= 2e-5, "colour differs on a surface without texels / cells"
    if loose.any():
        assert np.mean(dcol[loose] > 2e-5) <= 2e-3
    # a miss is all zeros apart from t = inf and the indices
    miss = ~hit
    assert not got["point"][miss].any() and not got["color"][miss].any()
    assert np.all(got["type_index"][miss] == 0xFFFFFFFF)
    in_medium = hit & same_id & medium
    if in_medium.any():   # transforms.rs:160-172: normal (1,0,0), front face, u = v = 0, point = ray.at(t)
        m = in_medium
        assert np.allclose(got["t"][m], ref["t"][m], rtol=2e-5, atol=0)
        assert np.allclose(got["point"][m], ref["point"][m], rtol=1e-4, atol=1e-3)
        assert np.array_equal(got["normal"][m], np.tile(F([1, 0, 0]), (m.sum(), 1)))
        assert np.all(got["front_face"][m] == 1) and not got["u"][m].any() and not got["v"][m].any()
        assert np.array_equal(bits(got["color"][m]), bits(ref["color"][m]))
    return got, ref, surf


def test_cornel_box_records(scenes):
    """Translate(Rotate(Box3d)) with the mixed-space face test (transforms.rs:15-24, 94-105), the FlipFace light
    (transforms.rs:199-201), rect (u, v) (objects.rs:206-228) and the glass sphere."""
    g, o = scenes("cornel_box")
    cam, _ = g.camera()
    rays = np.concatenate([R.camera_rays(cam, 200, 200), random_rays(60000, [1, 1, 1], [554, 554, 554], 21)])
    got, ref, surf = compare_records(g, o, rays)
    kinds = set(np.unique(got["material_kind"][surf]).tolist())
    assert kinds == {0, 2, 3}                      # Lambertian, Dielectric, DiffuseLight
    box = surf & (got["type_index"] == 4)          # Translate<Rotate<Box3d>> is the fifth concrete type added
    assert box.sum() > 5000
    # the rotated box shows both face flags (rays from inside the room hit it from outside; the mixed-space test
    # of Rotate flips some) and normals that are not axis-aligned
    assert 0 < got["front_face"][box].mean() <= 1
    assert np.any(np.abs(got["normal"][box][:, 0]) > 0.2) and np.any(np.abs(got["normal"][box][:, 2]) > 0.2)
    light = surf & (got["material_kind"] == 3)
    assert light.sum() > 100
    lit = got["color"][light].max(axis=1) > 0
    assert np.array_equal(lit, got["front_face"][light] == 1)      # material.rs:151-158 with FlipFace's flag


def test_final_scene_records(scenes):
    """Boxes, the moving sphere, glass, fuzzy metal, the two media, the earth (image texture addressed through the
    sphere's (u, v), texture.rs:101-122), the noise sphere (perlin.rs) and the instanced spheres behind
    Translate(Rotate(World))."""
    g, o = scenes("final_scene")
    cam, _ = g.camera()
    rays = np.concatenate([R.camera_rays(cam, 320, 180), random_rays(60000, [-200, 20, -200], [600, 540, 600], 22)])
    # concrete types in order of first add (final_scene.rs:49-168): 0 World of boxes, 1 FlipFace light, 2 moving
    # sphere, 3 glass, 4 metal, 5 ConstantMedium, 6 earth, 7 noise sphere, 8 Translate<Rotate<World>>
    got, ref, surf = compare_records(g, o, rays, medium_type=5, texel_types=(6,))
    for type_index, at_least in ((0, 20000), (2, 300), (4, 300), (6, 1000), (7, 1000), (8, 1000)):
        assert (surf & (got["type_index"] == type_index)).sum() >= at_least, type_index
    earth = surf & (got["type_index"] == 6)
    assert len(np.unique(got["color"][earth], axis=0)) > 200          # many different texels
    noise = surf & (got["type_index"] == 7)
    c = got["color"][noise]
    assert np.array_equal(c[:, 0], c[:, 1]) and c[:, 0].min() >= 0 and c[:, 0].max() <= 1 and c[:, 0].std() > 0.1
    inst = surf & (got["type_index"] == 8)
    assert len(np.unique(got["object_index"][inst])) == 1             # one top-level object, 1000 primitives


def test_spheres_records(scenes):
    """Moving spheres (centre at the ray's time in the normal, objects.rs:160-164), checker ground (texture.rs:52-59),
    metal and glass; through the BVH container."""
    g, o = scenes("spheres_bvh")
    cam, _ = g.camera()
    rays = np.concatenate([R.camera_rays(cam, 240, 135, time=0.75),
                           random_rays(60000, [-14, 0.0, -14], [14, 3, 14], 23)])
    got, ref, surf = compare_records(g, o, rays, texel_types=(0,))     # type 0: the checkered ground sphere
    assert set(np.unique(got["material_kind"][surf]).tolist()) == {0, 1, 2}
    inside = surf & (got["front_face"] == 0)
    assert inside.sum() > 100                                         # rays that start inside spheres


@pytest.mark.parametrize("pairs", [False, True])
@pytest.mark.parametrize("name,medium_type,texel_types,box,seed", [("cornel_box", None, (), ([1, 1, 1], [554, 554, 554]), 24),
                                                                   ("final_scene", 5, (6,), ([-200, 20, -200], [600, 540, 600]), 25)])
def test_records_of_small_scenes_walked_like_large_ones(name, medium_type, texel_types, box, seed, pairs, monkeypatch):
    """MIRT_NO_LDS keeps a scene out of LDS: its tree is folded into four-child nodes with quantised boxes and walked
    from global memory, its records, wrapper chains, materials and lights are read from there -- the code a scene of
    more than 4096 primitives runs, here with every kind of primitive, wrapper, medium and texture the examples
    have.  Same records as the oracle's, to the same bars, and a same-seed frame that matches per pixel.
    `pairs`: two primitives of a leaf per turn, as scenes from a quarter of a million primitives up are walked --
    here on the host builder's leaves of up to four, with rects, boxes, moving spheres and wrapped primitives going
    through the turn that fetches both records at once."""
    monkeypatch.setenv("MIRT_NO_LDS", "1")
    if pairs:
        monkeypatch.setenv("MIRT_PRIMS_PER_TURN", "2")
    g, o = R.ExampleScene(name), O.Scene(name)
    try:
        cam, _ = g.camera()
        rays = np.concatenate([R.camera_rays(cam, 160, 90), random_rays(30000, box[0], box[1], seed)])
        got, ref, surf = compare_records(g, o, rays, medium_type=medium_type, texel_types=texel_types)
        assert surf.sum() > 20000
        rgb, sums, st = g.render(96, 96, 4, seed=3)
        assert not st["scene_in_lds"]
        orgb, osums, _ = o.render(96, 96, 4, seed=3)
        ok = np.isfinite(osums).all(axis=2) & np.isfinite(sums).all(axis=2)
        close = np.isclose(sums, osums, rtol=1e-4, atol=1e-5).all(axis=2)
        assert np.mean(close[ok]) > 0.97               # (a libm rounding may flip a path)
    finally:
        g.close()
        o.close()
