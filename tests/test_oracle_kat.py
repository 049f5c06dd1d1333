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


# ---- wrappers (transforms.rs) ---------------------------------------------------------------------------------

def test_translate_recomputes_the_face_flag_against_the_forwarded_normal():
    # transforms.rs:15-24.  Sphere r=2 at the object-space origin, moved by (10,0,0); a ray from its centre along +x:
    # the sphere reports t=2, outward (1,0,0), back face, normal forwarded to (-1,0,0) (hittable.rs:28-35).
    # Translate then sets front_face = dot(direction, normal) < 0 with that forwarded normal: (1,0,0).(-1,0,0) < 0
    # -> TRUE, and leaves the normal alone.  (The bare sphere says back face: test_sphere_second_root_from_inside.)
    ok, out = O.kat_wrapped_sphere(1, [0, 0, 0, 2], [10, 0, 0, 1, 0, 0, 0], offset=(10, 0, 0))
    assert ok and out[0] == 2.0 and out[3] == 1.0
    assert tuple(out[4:7]) == (12.0, 0.0, 0.0)            # point moved back by the offset (:18)
    assert tuple(out[7:10]) == (-1.0, 0.0, 0.0)
    # from outside nothing changes: front face, outward normal
    ok, out = O.kat_wrapped_sphere(1, [0, 0, 0, 2], [5, 0, 0, 1, 0, 0, 0], offset=(10, 0, 0))
    assert ok and out[0] == 3.0 and out[3] == 1.0 and tuple(out[4:7]) == (8.0, 0.0, 0.0)
    assert tuple(out[7:10]) == (-1.0, 0.0, 0.0)


def test_rotate_tests_the_face_with_the_object_space_direction_against_the_world_space_normal():
    # transforms.rs:81-110 at 90 degrees (sin = 1, cos = cos_f32(pi/2) ~ -4.4e-8).  World -> object: x' = c x - s z,
    # z' = s x + c z (:85-89), so the world ray o=(2,0,-5), d=(-1,0,1) becomes o'=(5,0,2), d'=(-1,0,-1).  Against the
    # sphere at (2,0,0), r=1: os=(3,0,2), a=2, half_b=-5, c=12, disc=1, root=(5-1)/2=2 -> point (3,0,0), outward
    # (1,0,0), front (d'.n = -1).  Back to the world (:97-101): point (0,0,-3), normal (0,0,-1); the world ray hits
    # that side from outside (d.n = -1): a front face.  But the face test (:104) takes d' (object space) against
    # the world-space normal: (-1,0,-1).(0,0,-1) = +1, not < 0, so the hit is reported as a BACK face and the
    # normal is negated to (0,0,+1).
    ok, out = O.kat_wrapped_sphere(2, [2, 0, 0, 1], [2, 0, -5, -1, 0, 1, 0], angle=90.0)
    assert ok and abs(out[0] - 2.0) < 1e-6
    assert out[3] == 0.0
    assert np.allclose(out[4:7], [0, 0, -3], atol=1e-6)
    assert np.allclose(out[7:10], [0, 0, 1], atol=1e-6)
    # at 0 degrees Rotate is the identity and the flag is the sphere's own
    ok, out = O.kat_wrapped_sphere(2, [0, 0, -3, 1], [0, 0, -8, 0, 0, 1, 0], angle=0.0)
    assert ok and out[0] == 4.0 and out[3] == 1.0 and tuple(out[7:10]) == (0.0, 0.0, -1.0)
    # 180 degrees: d' = -d, the normal turns with it, dot(d', n_world) = dot(-d, n_world) > 0 for a front hit:
    # every front hit is reported as a back face with the normal flipped along the ray
    ok, out = O.kat_wrapped_sphere(2, [0, 0, 3, 1], [0, 0, -8, 0, 0, 1, 0], angle=180.0)   # world centre (0,0,-3)
    assert ok and abs(out[0] - 4.0) < 1e-5 and out[3] == 0.0 and np.allclose(out[7:10], [0, 0, 1], atol=1e-6)


def test_translate_of_rotate_as_the_cornell_box_nests_them():
    # cornel_box.rs:90-99: Translate(Rotate(obj, angle), offset).  The case above moved by (0,0,10): Translate
    # hands Rotate the ray o - offset; Rotate reports back face, normal (0,0,1); Translate adds the offset to the
    # point and recomputes the flag with the world direction against that normal: (-1,0,1).(0,0,1) > 0 -> back face
    # stays, normal untouched.
    ok, out = O.kat_wrapped_sphere(4, [2, 0, 0, 1], [2, 0, 5, -1, 0, 1, 0], offset=(0, 0, 10), angle=90.0)
    assert ok and abs(out[0] - 2.0) < 1e-5 and out[3] == 0.0
    assert np.allclose(out[4:7], [0, 0, 7], atol=1e-5) and np.allclose(out[7:10], [0, 0, 1], atol=1e-6)
    # 180 degrees inside a Translate: Rotate says back face with the normal along the ray (previous test), and
    # Translate's own test against that normal, (0,0,1).(0,0,1) > 0, agrees
    ok, out = O.kat_wrapped_sphere(4, [0, 0, 3, 1], [5, 0, -8, 0, 0, 1, 0], offset=(5, 0, 0), angle=180.0)
    assert ok and abs(out[0] - 4.0) < 1e-5 and out[3] == 0.0 and np.allclose(out[7:10], [0, 0, 1], atol=1e-6)


def test_flip_face_toggles_the_flag_only():
    # transforms.rs:197-205
    ok, out = O.kat_wrapped_sphere(3, [0, 0, 0, 1], [0, 0, -5, 0, 0, 1, 0])
    assert ok and out[0] == 4.0 and out[3] == 0.0 and tuple(out[7:10]) == (0.0, 0.0, -1.0)


# ---- constant medium (transforms.rs:133-180) -----------------------------------------------------------------

def media_draw(seed, pixel):
    """Draw 0 of the media stream of (pixel, sample 0, bounce 0): Philox counter (pixel, 0, stage 1, block 0)."""
    word = O.philox([pixel, 0, 1, 0], [seed & 0xFFFFFFFF, seed >> 32])[0]
    return F(word >> 8) * F(1.0 / 16777216.0)


def test_constant_medium_free_flight():
    # Unit sphere at the origin, ray o=(0,0,-5), d=(0,0,2): boundary hits at t=2 and t=3 (:135-136), |d|=2, so
    # distance_inside = (3-2)*2 = 2 (:152-153).  hit_distance = -1/density * log2(xi) (:154 -- log2, not ln);
    # accepted when <= 2, then t = 2 + hit_distance/2, normal (1,0,0), front face, u=v=0 (:160-172).
    sphere, ray = [0, 0, 0, 1], [0, 0, -5, 0, 0, 2, 0]
    hits = misses = 0
    for pixel in range(40):
        xi = media_draw(9, pixel)
        density = F(0.8)
        neg_inv = F(-1.0) / density
        hit_distance = neg_inv * F(math.log2(float(xi))) if xi > 0 else F(np.inf)
        ok, out = O.kat_medium(sphere, float(density), ray, seed=9, pixel=pixel)
        assert ok == bool(hit_distance <= F(2.0))
        if ok:
            hits += 1
            assert abs(out[0] - (2.0 + float(hit_distance) / 2.0)) < 2e-6
            assert tuple(out[7:10]) == (1.0, 0.0, 0.0) and out[3] == 1.0 and out[1] == 0.0 and out[2] == 0.0
            assert np.allclose(out[4:7], [0, 0, -5 + 2 * out[0]], atol=1e-5)
        else:
            misses += 1
    assert hits > 5 and misses > 5


def test_constant_medium_clamps_to_the_callers_range():
    sphere = [0, 0, 0, 1]
    # t_max in front of the boundary's entry: hit1.t=2 >= hit2.t=min(3, 1.5) -> none (:140-146)
    for pixel in range(8):
        ok, _ = O.kat_medium(sphere, 1e6, [0, 0, -5, 0, 0, 2, 0], seed=1, pixel=pixel, t_max=1.5)
        assert not ok
    # origin inside: hit1.t = -0.5 is raised to t_min = 0.001 (:137-139); a very dense medium scatters at once
    ok, out = O.kat_medium(sphere, 1e9, [0, 0, 0, 0, 0, 2, 0], seed=1, pixel=0)
    assert ok and abs(out[0] - 0.001) < 1e-6
    # a ray that misses the boundary
    ok, _ = O.kat_medium(sphere, 1e9, [0, 3, -5, 0, 0, 2, 0], seed=1, pixel=0)
    assert not ok


def test_constant_medium_negative_zero_density():
    # ConstantMedium::new: neg_inv_density = -1.0 / -0.0 = +inf (:128); hit_distance = +inf * log2(xi) = -inf for
    # xi in (0,1); `-inf > distance_inside` is false, so the medium always scatters, at t = hit1.t + (-inf)/|d| = -inf
    ok, out = O.kat_medium([0, 0, 0, 1], -0.0, [0, 0, -5, 0, 0, 2, 0], seed=3, pixel=1)
    assert 0 < media_draw(3, 1) < 1
    assert ok and out[0] == -np.inf and out[3] == 1.0


# ---- light pdfs (objects.rs:82-94, 292-301; pdf.rs:81-83) ---------------------------------------------------------

def test_xz_rect_pdf_value_closed_form():
    rect = [0, 2, 0, 4, 1]                      # area 8, plane y = 1
    # straight up from (1,0,2) with |d| = 2: t = 0.5, t^2 |d|^2 = 1, cosine 1 -> 1/8
    assert O.kat_pdf_value(2, rect, [1, 0, 2], [0, 2, 0]) == 0.125
    # oblique d = (0,1,1) from (1,0,1): t = 1, hits (1,1,2); t^2 |d|^2 = 2, cosine = 1/sqrt 2 -> 2 sqrt 2 / 8
    assert abs(O.kat_pdf_value(2, rect, [1, 0, 1], [0, 1, 1]) - 2 * math.sqrt(2) / 8) < 1e-7
    # pointing away / missing the rect / starting on it (t < 0.0001)
    assert O.kat_pdf_value(2, rect, [1, 0, 2], [0, -1, 0]) == 0.0
    assert O.kat_pdf_value(2, rect, [5, 0, 2], [0, 1, 0]) == 0.0
    assert O.kat_pdf_value(2, rect, [1, 1, 2], [0, 1, 0]) == 0.0
    # the edge still counts (strict comparisons, objects.rs:268-271)
    assert O.kat_pdf_value(2, rect, [2, 0, 4], [0, 1, 0]) == 0.125
    # XYRect keeps the trait default (hittable.rs:52-54)
    assert O.kat_pdf_value(1, rect, [1, 2, 0], [0, 0, 1]) == 0.0


def test_sphere_pdf_value_closed_form():
    # centre (0,0,3), r = 1 seen from the origin: cos_max = sqrt(1 - 1/9), pdf = 1 / (2 pi (1 - cos_max))
    expect = 1.0 / (2 * math.pi * (1 - math.sqrt(1 - 1 / 9)))
    assert abs(O.kat_pdf_value(0, [0, 0, 3, 1], [0, 0, 0], [0, 0, 1]) - expect) < 2e-6
    assert abs(O.kat_pdf_value(0, [0, 0, 3, 1], [0, 0, 0], [0, 0, 5]) - expect) < 2e-6      # any length
    assert O.kat_pdf_value(0, [0, 0, 3, 1], [0, 0, 0], [0, 1, 0]) == 0.0                     # misses the sphere
    # from inside the sphere: 1 - r^2/d^2 < 0 -> NaN, as the reference computes it
    assert math.isnan(O.kat_pdf_value(0, [0, 0, 3, 1], [0, 0, 3.5], [0, 0, 1]))


def test_mixture_pdf_value_is_the_even_average():
    rect = [0, 2, 0, 4, 1]
    # light pdf 1/8 (above), cosine pdf cos/pi = 1/pi around the normal (0,1,0) -> (1/8 + 1/pi)/2
    got = O.kat_mixture_value(rect, [1, 0, 2], [0, 1, 0], [0, 2, 0])
    assert abs(got - 0.5 * (0.125 + 1 / math.pi)) < 1e-7
    # below the horizon of the normal the cosine term is 0 (pdf.rs:25-32), the light term stays
    got = O.kat_mixture_value(rect, [1, 0, 2], [0, -1, 0], [0, 2, 0])
    assert abs(got - 0.0625) < 1e-7
    # a direction that misses the light: cosine term only (45 degrees: cos = 1/sqrt 2)
    got = O.kat_mixture_value(rect, [1.5, 0, 2], [0, 1, 0], [9, 9, 0])      # crosses y = 1 at x = 2.5
    assert abs(got - 0.5 * (1 / math.sqrt(2)) / math.pi) < 1e-7


# ---- Perlin noise with tables that can be followed by hand (perlin.rs:26-60, 75-93) ------------------------------

def test_perlin_noise_and_turbulence_with_identity_tables():
    ident = np.arange(256, dtype=np.uint32)
    vec = np.zeros((256, 3), F)
    vec[:, 0] = np.arange(256)                # gradient h = (h, 0, 0): the value names the lattice hash
    # p = (3.25, 5, 9): i,j,k = 3,5,9; u = .25, v = w = 0 -> vv = ww = 0, only dj = dk = 0 contribute.
    # uu = .25^2 (3 - .5) = .15625.  Hashes 3^5^9 = 15 and 4^5^9 = 8.
    # noise = (1-uu) * 15 * u + uu * 8 * (u - 1) = .84375 * 3.75 - .15625 * 6 = 2.2265625   (exact in binary)
    assert O.kat_perlin(vec, ident, ident, ident, [3.25, 5, 9]) == 2.2265625
    # turb depth 2 adds 0.5 * noise(2p), 2p = (6.5, 10, 18): u = .5, uu = .5, hashes 6^10^18 = 30 and 7^10^18 = 31:
    # .5 * 30 * .5 + .5 * 31 * (-.5) = -0.25 ; |2.2265625 - 0.125| = 2.1015625
    assert O.kat_perlin(vec, ident, ident, ident, [6.5, 10, 18]) == -0.25
    assert O.kat_perlin(vec, ident, ident, ident, [3.25, 5, 9], depth=2) == 2.1015625
    # negative coordinates: floor(-0.75) = -1, (-1 & 255) = 255 and (0 & 255) = 0, u = .25
    assert O.kat_perlin(vec, ident, ident, ident, [-0.75, 0, 0]) == 0.84375 * 255 * 0.25
    # the three tables are combined by XOR, each indexed by its own axis
    px = (ident + 1) & 255
    # i = 3 -> perm_x[3] = 4, perm_x[4] = 5: hashes 4^5^9 = 8 and 5^5^9 = 9
    assert O.kat_perlin(vec, px, ident, ident, [3.25, 5, 9]) == 0.84375 * 8 * 0.25 + 0.15625 * 9 * (-0.75)
    # a gradient along y only shows through v
    vy = np.zeros((256, 3), F)
    vy[:, 1] = 1.0
    assert O.kat_perlin(vy, ident, ident, ident, [3.25, 5, 9]) == 0.0
    # v = .5: vv = .5 -> .5 * .5 + .5 * (-.5) = 0 ; v = .25 -> .84375 * .25 - .15625 * .75 = 0.09375
    assert O.kat_perlin(vy, ident, ident, ident, [3, 5.25, 9]) == 0.09375


# ---- image texture addressing (texture.rs:101-122), sphere (u, v) (objects.rs:17-28), checker (texture.rs:51-60) ----

def test_image_texture_addressing():
    img = np.zeros((2, 4, 3), np.uint8)        # 4 wide, 2 high; texel (i, j) = (10 i, 100 j, 255)
    for j in range(2):
        for i in range(4):
            img[j, i] = (10 * i, 100 * j, 255)

    def texel(u, v):
        c = O.kat_image_texel(img, u, v)
        assert c[2] == 1.0
        return round(float(c[0]) * 255 / 10), round(float(c[1]) * 255 / 100)
    assert texel(0.0, 1.0) == (0, 0)           # v is flipped: v = 1 is row 0, the top of the image
    assert texel(0.0, 0.0) == (0, 1)           # v = 0 -> 1 * height = 2 -> clamped to the last row
    assert texel(1.0, 1.0) == (3, 0)           # u = 1 -> 4 -> clamped to the last column
    assert texel(0.5, 0.5) == (2, 1)           # .5 * 4 = 2 ; (1 - .5) * 2 = 1
    assert texel(0.49, 0.51) == (1, 0)
    assert texel(-3.0, 7.0) == (0, 0)          # clamped to [0,1] first
    assert texel(9.0, -2.0) == (3, 1)
    assert texel(float("nan"), 1.0) == (0, 0)  # f32::clamp keeps NaN; NaN as u32 = 0
    c = O.kat_image_texel(img, 0.5, 1.0)
    assert c[0] == F(1.0 / 255.0) * F(20.0)    # COLOR_SCALE * byte, in f32


def test_sphere_uv_at_the_axes():
    # u = (-atan2(z, x) + pi) / 2 pi, v = acos(-y) / pi
    for p, u, v in (((1, 0, 0), 0.5, 0.5), ((-1, 0, 0), 0.0, 0.5), ((0, 0, 1), 0.25, 0.5), ((0, 0, -1), 0.75, 0.5),
                    ((0, 1, 0), 0.5, 1.0), ((0, -1, 0), 0.5, 0.0)):
        gu, gv = O.kat_sphere_uv(p)
        assert abs(gu - u) < 1e-7 and abs(gv - v) < 1e-7, p
    gu, gv = O.kat_sphere_uv([0.6, 0.0, 0.8])
    assert abs(gu - (math.pi - math.atan2(0.8, 0.6)) / (2 * math.pi)) < 1e-7 and abs(gv - 0.5) < 1e-7


def test_checker_sign_of_the_product_of_sines():
    # sin(10x) sin(10y) sin(10z) < 0 -> odd.  10 * 0.1 = 1 rad: all positive -> even; one negative coordinate -> odd
    assert not O.kat_checker_is_odd([0.1, 0.1, 0.1])
    assert O.kat_checker_is_odd([-0.1, 0.1, 0.1])
    assert not O.kat_checker_is_odd([-0.1, -0.1, 0.1])
    assert O.kat_checker_is_odd([0.1, 0.1, 0.4])      # sin(4) < 0
    assert not O.kat_checker_is_odd([0.0, 0.1, 0.1])  # product 0 is not < 0


@pytest.mark.parametrize("name,look_from,look_at,aspect,aperture", [
    ("spheres", (13.0, 2.0, 3.0), (0.0, 0.0, 0.0), 16.0 / 9.0, 0.1),          # examples/spheres.rs:3,12-28
    ("spheres_bvh", (13.0, 2.0, 3.0), (0.0, 0.0, 0.0), 16.0 / 9.0, 0.1),      # examples/spheres_bvh.rs:3,14-30
    ("cornel_box", (478.0, 278.0, -600.0), (278.0, 278.0, 0.0), 1.0, 0.0),    # examples/cornel_box.rs:3,28-44
    ("final_scene", (478.0, 278.0, -600.0), (278.0, 278.0, 0.0), 1.0, 0.0)])  # examples/final_scene.rs:4,18-34
def test_camera_new_against_the_formulas_in_double_precision(name, look_from, look_at, aspect, aperture):
    """Camera::new (camera.rs:30-56) for the four example programs: vfov 40, v_up (0,1,0), focus distance 10, shutter
    [0, 1].  Expected values worked out here in float64 from the same formulas; the oracle computes in f32."""
    cam, _ = O.Scene(name).camera()
    lf, la, vup = np.array(look_from), np.array(look_at), np.array([0.0, 1.0, 0.0])
    h = math.tan(math.radians(40.0) / 2.0)
    vh, focus = 2.0 * h, 10.0
    vw = aspect * vh
    w = (lf - la) / np.linalg.norm(lf - la)
    u = np.cross(vup, w); u /= np.linalg.norm(u)
    v = np.cross(w, u)
    hor, ver = focus * vw * u, focus * vh * v
    llc = lf - 0.5 * hor - 0.5 * ver - focus * w
    want = np.concatenate([lf, llc, hor, ver, u, v, w, [aperture / 2.0, 0.0, 1.0]])
    assert cam.shape == (24,)
    assert np.allclose(cam, want, rtol=2e-6, atol=2e-5), (cam, want)
    # a ray through the middle of the image points at look_at (camera.rs:61-62 with no lens offset)
    mid = (cam[3:6] + 0.5 * cam[6:9] + 0.5 * cam[9:12]) - cam[0:3]
    assert np.allclose(mid / np.linalg.norm(mid), (la - lf) / np.linalg.norm(la - lf), atol=1e-5)


def _shading_draws(seed, n=2):
    """The first draws of the shading stage of path (pixel 0, sample 0, bounce 0): Philox4x32-10 on counter
    (0, 0, 4 * 0 + 2, 0), each word's upper 24 bits scaled to [0, 1)."""
    words = O.philox([0, 0, 2, 0], [seed & 0xFFFFFFFF, seed >> 32])
    return [F(w >> 8) * F(1.0 / 16777216.0) for w in words[:n]]


@pytest.mark.parametrize("seed", [1, 2, 77])
def test_random_cosine_direction_from_its_two_draws(seed):
    """vec3.rs:62-74: z = sqrt(1 - r2), (x, y) = (cos, sin)(2 pi r1) sqrt(r2): a unit vector in the upper hemisphere."""
    r1, r2 = _shading_draws(seed)
    d = O.kat_direction(0, seed)
    want = np.array([math.cos(2 * math.pi * r1) * math.sqrt(r2), math.sin(2 * math.pi * r1) * math.sqrt(r2),
                     math.sqrt(1.0 - r2)])
    assert np.allclose(d, want, atol=2e-6)
    assert abs(float(np.linalg.norm(d.astype(np.float64))) - 1.0) < 1e-6 and d[2] >= 0


@pytest.mark.parametrize("seed,radius,dist2", [(3, 1.0, 4.0), (4, 90.0, 250000.0), (5, 0.5, 0.2500001)])
def test_random_to_sphere_stays_inside_the_cone_of_the_sphere(seed, radius, dist2):
    """vec3.rs:76-88: z = 1 + r2 (cos_theta_max - 1) with cos_theta_max = sqrt(1 - R^2 / d^2), so z runs from 1 (the
    centre) down to cos_theta_max (the rim); (x, y) on the circle of radius sqrt(1 - z^2)."""
    r1, r2 = _shading_draws(seed)
    d = O.kat_direction(1, seed, radius, dist2)
    cos_max = math.sqrt(max(0.0, 1.0 - float(F(radius) * F(radius) / F(dist2))))
    z = 1.0 + r2 * (cos_max - 1.0)
    s = math.sqrt(max(0.0, 1.0 - z * z))
    want = np.array([math.cos(2 * math.pi * r1) * s, math.sin(2 * math.pi * r1) * s, z])
    assert np.allclose(d, want, atol=5e-4 if dist2 < 1 else 2e-6)      # (at the rim of a sphere seen from its surface 1 - z^2 cancels)
    assert cos_max - 1e-6 <= d[2] <= 1.0 + 1e-6


def test_schlick_reflectance_known_values():
    """material.rs:105-108: r0 = ((1 - n) / (1 + n))^2, r0 + (1 - r0) (1 - cos)^5."""
    assert O.kat_reflectance(1.0, 1.5) == pytest.approx(0.04, abs=1e-7)             # head on: r0 = (0.5 / 2.5)^2
    assert O.kat_reflectance(0.0, 1.5) == pytest.approx(1.0, abs=1e-7)              # grazing: everything reflects
    assert O.kat_reflectance(0.5, 1.5) == pytest.approx(0.04 + 0.96 / 32.0, abs=1e-7)
    assert O.kat_reflectance(1.0, 1.0 / 1.5) == pytest.approx(0.04, abs=1e-7)       # the same from the inside


@pytest.mark.parametrize("with_sphere", [False, True])
def test_light_sampling_keeps_a_furnace_in_balance(with_sphere):
    """renderer.rs:166-183 with the mixture of pdf.rs:81-91: under a sky and lights of one radiance b the floor's
    expected radiance is albedo * b whichever way directions are drawn, provided the densities of
    objects.rs:82-100, 292-309 and world.rs:100-117 match the way they are drawn."""
    got = O.kat_furnace(with_sphere, 0.5, 0.5, 200000, seed=11)
    assert abs(got - 0.25) < 0.0025, got


def test_culling_container_equals_the_linear_scan():
    """oracle.hpp's CulledBVH is not in the reference: it stands in for the reference's exhaustive containers where
    those would take hours (the CPU baseline and the checker of ten-million-sphere scenes).  It must return what
    World::hit's scan returns (world.rs:75-94), ids and t bits, also for rays that skim a field for a thousand
    units, where the f32 sphere test reports hits far off the spheres."""
    half = 150
    a, c = O.Scene(f"synthetic:{half}"), O.Scene(f"synthetic_culled:{half}")
    assert a.object_count() == c.object_count() > 90000
    rng = np.random.default_rng(12)
    n = 600
    rays = np.zeros((n, 7), np.float32)
    angle = rng.uniform(0, 2 * np.pi, n)
    start = rng.uniform(0.95, 1.3, n) * half
    rays[:, 0] = -np.cos(angle) * start
    rays[:, 2] = -np.sin(angle) * start
    rays[:, 1] = rng.uniform(0.45, 3.0, n)
    reach = rng.uniform(0.8, 2.6, n) * half
    scale = rng.uniform(0.2, 3.0, n)
    rays[:, 3] = np.cos(angle) * scale
    rays[:, 5] = np.sin(angle) * scale
    rays[:, 4] = -(rays[:, 1] - 0.4) / reach * scale
    rays[:, 6] = rng.uniform(0, 1, n)
    cam, _ = a.camera()
    rays = np.concatenate([rays, O.pixel_centre_rays(cam, 48, 27)])
    ti, oi, t = a.trace(rays)
    cti, coi, ct = c.trace(rays)
    assert np.array_equal(ti, cti) and np.array_equal(oi, coi)
    assert np.array_equal(t.view(np.uint32), ct.view(np.uint32))
    assert (ti != 0xFFFFFFFF).sum() > 800
