"""Known answers for the CPU oracle, derivable by hand from the reference sources (file:line in each test).

The reference has no tests for this path (its only tests are blobvec.rs:85-159), so these
anchor the oracle on the source semantics themselves.
"""
import math

import numpy as np
import pytest

import oracle_lib as O

F = np.float32


def test_philox_known_answers():
    # Philox4x32-10 test vectors (Random123 kat_vectors)
    assert O.philox([0, 0, 0, 0], [0, 0]) == [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]
    assert O.philox([0xFFFFFFFF] * 4, [0xFFFFFFFF] * 2) == [0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD]
    assert O.philox([0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344], [0xA4093822, 0x299F31D0]) == \
        [0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1]


def test_sphere_hit_unit_sphere():
    # objects.rs:41-73: unit sphere at the origin, ray o=(0,0,-5) d=(0,0,1): a=1, half_b=-5, c=24, disc=1, root=4
    ok, out = O.kat_hit(0, [0, 0, 0, 1], [0, 0, -5, 0, 0, 1, 0])
    assert ok
    t, u, v, front = out[0:4]
    assert t == 4.0
    assert front == 1.0
    assert tuple(out[4:7]) == (0.0, 0.0, -1.0)      # point
    assert tuple(out[7:10]) == (0.0, 0.0, -1.0)     # outward normal, front face
    # objects.rs:17-28: theta = acos(-0) = pi/2 -> v = 0.5 ; phi = -atan2(-1, 0) + pi = 3pi/2 -> u = 0.75
    assert abs(v - 0.5) < 1e-7 and abs(u - 0.75) < 1e-7


def test_sphere_second_root_from_inside():
    # origin inside: first root negative (< t_min) so the far root is taken, normal flipped (hittable.rs:28-35)
    ok, out = O.kat_hit(0, [0, 0, 0, 2], [0, 0, 0, 1, 0, 0, 0])
    assert ok and out[0] == 2.0 and out[3] == 0.0
    assert tuple(out[7:10]) == (-1.0, 0.0, 0.0)


def test_sphere_miss_and_range():
    ok, _ = O.kat_hit(0, [0, 0, 0, 1], [0, 2, -5, 0, 0, 1, 0])
    assert not ok
    # both roots above t_max
    ok, _ = O.kat_hit(0, [0, 0, 0, 1], [0, 0, -5, 0, 0, 1, 0], t_min=0.001, t_max=3.0)
    assert not ok
    # t == t_max is still a hit: the rejection is `root > t_max` (objects.rs:54)
    ok, out = O.kat_hit(0, [0, 0, 0, 1], [0, 0, -5, 0, 0, 1, 0], t_min=0.001, t_max=4.0)
    assert ok and out[0] == 4.0


def test_sphere_unnormalised_direction_scales_t():
    # camera.rs:61-62 never normalises: with d=(0,0,2) the same hit is at t=2
    ok, out = O.kat_hit(0, [0, 0, 0, 1], [0, 0, -5, 0, 0, 2, 0])
    assert ok and out[0] == 2.0


def test_moving_sphere_centre_follows_time():
    # objects.rs:132-135: centre(t) = c0 + ((t - t0)/(t1 - t0)) * (c1 - c0); at time 0.5 centre = (0, 1, 0)
    p = [0, 0, 0, 0, 2, 0, 0.0, 1.0, 1.0]
    ok, out = O.kat_hit(5, p, [0, 1, -5, 0, 0, 1, 0.5])
    assert ok and out[0] == 4.0
    ok, _ = O.kat_hit(5, p, [0, 3.5, -5, 0, 0, 1, 0.5])
    assert not ok


def test_xz_rect_edges_hit_and_uv():
    # objects.rs:261-283: strict comparisons, so a ray through the edge x == x1 hits; u = (x-x0)/(x1-x0)
    rect = [0, 2, 0, 4, 1]  # x0,x1,z0,z1,k
    ok, out = O.kat_hit(2, rect, [2, 0, 1, 0, 1, 0, 0])
    assert ok and out[0] == 1.0 and out[1] == 1.0 and out[2] == 0.25
    assert tuple(out[7:10]) == (0.0, -1.0, 0.0) and out[3] == 0.0   # hit from below: back face
    ok, _ = O.kat_hit(2, rect, [2.0001, 0, 1, 0, 1, 0, 0])
    assert not ok
    # parallel ray: t = inf or NaN is rejected by the range / bounds tests
    ok, _ = O.kat_hit(2, rect, [1, 0, 1, 1, 0, 0, 0])
    assert not ok


def test_xy_yz_rect_normals():
    ok, out = O.kat_hit(1, [0, 1, 0, 1, 3], [0.5, 0.5, 0, 0, 0, 1, 0])
    assert ok and out[0] == 3.0 and tuple(out[7:10]) == (0.0, 0.0, -1.0)
    ok, out = O.kat_hit(3, [0, 1, 0, 1, -2], [0, 0.5, 0.25, -1, 0, 0, 0])
    assert ok and out[0] == 2.0 and tuple(out[7:10]) == (1.0, 0.0, 0.0) and out[3] == 1.0
    assert out[1] == 0.5 and out[2] == 0.25


def test_box_closest_face_and_order():
    # objects.rs:396-436: closest of six rects
    box = [0, 0, 0, 1, 2, 3]
    ok, out = O.kat_hit(4, box, [0.5, 1, -1, 0, 0, 1, 0])
    assert ok and out[0] == 1.0 and tuple(out[7:10]) == (0.0, 0.0, -1.0)
    ok, out = O.kat_hit(4, box, [0.5, 5, 1, 0, -1, 0, 0])
    assert ok and out[0] == 3.0 and tuple(out[7:10]) == (0.0, 1.0, 0.0)
    # from inside the box the exit face is found
    ok, out = O.kat_hit(4, box, [0.5, 1, 1, 1, 0, 0, 0])
    assert ok and out[0] == 0.5


def test_onb_branches():
    # onb.rs:10-20: |w.x| > 0.9 picks a = (0,1,0), else (1,0,0)
    b = O.kat_onb([1, 0, 0])
    assert tuple(b[2]) == (1.0, 0.0, 0.0)
    assert tuple(b[1]) == (0.0, 0.0, 1.0)      # v = unit(w x (0,1,0)) = (0,0,1)
    assert tuple(b[0]) == (0.0, -1.0, 0.0)     # u = w x v
    b = O.kat_onb([0, 0, 2])
    assert tuple(b[2]) == (0.0, 0.0, 1.0)
    assert tuple(b[1]) == (0.0, 1.0, 0.0)      # v = unit(w x (1,0,0)) = (0,1,0)
    assert tuple(b[0]) == (-1.0, 0.0, 0.0)
    for n in ([0.3, -0.2, 0.9], [0.95, 0.1, 0.1]):
        b = O.kat_onb(n).astype(np.float64)
        assert np.allclose(b @ b.T, np.eye(3), atol=1e-6)


def test_dielectric_total_internal_reflection():
    # material.rs:111-137: leaving glass (ior 1.5, back face) at a grazing angle: ratio*sin > 1 -> reflect, no draw
    d = np.array([1.0, -0.2, 0.0], F)
    d /= np.linalg.norm(d)
    n = np.array([0, 1, 0], F)
    out = O.kat_dielectric(1.5, d, n, front_face=False)
    assert np.allclose(out, [d[0], -d[1], 0], atol=1e-6)


def test_dielectric_head_on_refracts_straight():
    # cos = 1: reflectance = r0 = ((1-1/1.5)/(1+1/1.5))^2 = 0.04; almost every draw refracts straight through
    n = np.array([0, 0, -1], F)
    straight = 0
    for seed in range(64):
        out = O.kat_dielectric(1.5, [0, 0, 1], n, front_face=True, seed=seed)
        if np.allclose(out, [0, 0, 1], atol=1e-6):
            straight += 1
        else:
            assert np.allclose(out, [0, 0, -1], atol=1e-6)
    assert straight >= 56


def test_encode_channel():
    # renderer.rs:241-252: (255.999 * sqrt(c/spp).clamp(0, 0.999)) as u8 ; NaN -> 0
    enc = O.lib().oracle_kat_encode
    assert enc(0.0, 1) == 0
    assert enc(1.0, 1) == 255           # 255.999 * 0.999 = 255.74 -> 255
    assert enc(100.0, 1) == 255
    assert enc(0.25, 1) == 127          # sqrt = 0.5 -> 127.9995 -> 127
    assert enc(4.0, 16) == 127
    assert enc(float("nan"), 4) == 0
    assert enc(-1.0, 4) == 0            # sqrt of a negative is NaN
    assert enc(float("inf"), 4) == 255


def test_aabb_hit_is_always_true_on_this_path():
    # aabb.rs:41-51 widens the interval, so with t_max = inf the test cannot fail, even for a ray pointing away
    rng = np.random.default_rng(5)
    hit = O.lib().oracle_kat_aabb_hit
    for _ in range(200):
        box = np.sort(rng.uniform(-5, 5, (2, 3)), axis=0).astype(F).ravel()
        ray = np.concatenate([rng.uniform(-20, 20, 3), rng.uniform(-1, 1, 3), [0.0]]).astype(F)
        assert hit(box.ctypes.data, ray.ctypes.data, 0.001, float("inf")) == 1
    box = np.array([0, 0, 0, 1, 1, 1], F)
    ray = np.array([5, 5, 5, 1, 1, 1, 0], F)
    assert hit(box.ctypes.data, ray.ctypes.data, 0.001, float("inf")) == 1


def test_scene_object_counts():
    # examples/spheres.rs:49-104: ground + up to 484 small spheres + 3 big ones
    s = O.Scene("spheres")
    assert 400 < s.object_count() <= 488
    assert (s.width, s.height, s.spp, s.depth) == (600, 337, 10, 10)
    assert O.Scene("cornel_box").object_count() == 8          # examples/cornel_box.rs:66-102
    f = O.Scene("final_scene")
    assert f.object_count() == 10                            # examples/final_scene.rs:76-168
    assert (f.width, f.height, f.spp, f.depth) == (600, 600, 5, 5)


def test_scene_seed_changes_the_scene_only_when_it_should():
    a = O.Scene("spheres", scene_seed=1)
    b = O.Scene("spheres", scene_seed=1)
    c = O.Scene("spheres", scene_seed=2)
    cam, _ = a.camera()
    rays = O.pixel_centre_rays(cam, 64, 36)
    ra, rb, rc = a.trace(rays), b.trace(rays), c.trace(rays)
    assert all(np.array_equal(x, y) for x, y in zip(ra, rb))
    assert not np.array_equal(ra[2], rc[2])


def test_bvh_tournament_equals_exhaustive_scan():
    # bvh.rs:92-152 with the always-true AABB::hit is a closest-hit over all objects; same (id, t) as world.rs:75-94
    w = O.Scene("spheres", scene_seed=3)
    b = O.Scene("spheres_bvh", scene_seed=3)
    cam, _ = w.camera()
    rays = O.pixel_centre_rays(cam, 120, 68)
    rng = np.random.default_rng(11)
    extra = np.zeros((4000, 7), F)
    extra[:, 0:3] = rng.uniform(-12, 12, (4000, 3)) * [1, 0.2, 1] + [0, 1.0, 0]
    extra[:, 3:6] = rng.normal(size=(4000, 3))
    extra[:, 6] = rng.uniform(0, 1, 4000)
    rays = np.concatenate([rays, extra])
    tw, ow, tt = w.trace(rays)
    tb, ob, tbv = b.trace(rays)
    assert np.array_equal(tw, tb) and np.array_equal(ow, ob)
    assert np.array_equal(tt.view(np.uint32), tbv.view(np.uint32))


def test_render_is_deterministic_and_shards_by_sample_range():
    s = O.Scene("cornel_box")
    a = s.render(24, 24, 6, seed=5)[1]
    b = s.render(24, 24, 6, seed=5, threads=3)[1]
    assert np.array_equal(a, b)
    lo = s.render(24, 24, 4, seed=5, first_sample=0)[1]
    hi = s.render(24, 24, 2, seed=5, first_sample=4)[1]
    ok = np.isfinite(a)
    assert np.allclose((lo + hi)[ok], a[ok], rtol=1e-5, atol=1e-6)
    c = s.render(24, 24, 6, seed=6)[1]
    assert not np.array_equal(a, c)


def test_max_depth_zero_is_black_and_miss_is_background():
    s = O.Scene("spheres")
    rgb, sums, seg = s.render(16, 9, 2, depth=0)
    assert seg == 0 and not sums.any() and not rgb.any()
    # a ray into the sky returns the background (renderer.rs:187-189): top row of the image, depth 1
    rgb, sums, _ = s.render(32, 18, 4, depth=1)
    assert np.allclose(sums[0] / 4, 0.8, atol=1e-6)
