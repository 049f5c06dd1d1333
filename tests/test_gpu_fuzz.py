"""Random sphere fields through the C ABI against an exhaustive NumPy scan in the reference's f32 operation order.

The device walks a BVH with a padded (conservative) box test; the reference tests every object
(world.rs:75-94, and aabb.rs:41-51 never rejects).  Whatever the scale of the scene, the ids and the bits of t must
be those of the exhaustive scan, ties going to the later object."""
import ctypes as C

import numpy as np
import pytest

import raytracing_amd as R
from test_gpu_cabi import Ctx, sphere

pytestmark = pytest.mark.gpu
F = np.float32


def scan(centers, radii, rays, t_min=F(0.001)):
    """World::hit over spheres in insertion order (objects.rs:42-59), vectorised over rays; f32 throughout."""
    o, d = rays[:, 0:3].astype(F), rays[:, 3:6].astype(F)
    best_t = np.full(len(rays), np.inf, F)
    best_i = np.full(len(rays), 0xFFFFFFFF, np.uint32)
    a = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]
    with np.errstate(all="ignore"):
        for i, (c, r) in enumerate(zip(centers.astype(F), radii.astype(F))):
            os = o - c
            half_b = (os[:, 0] * d[:, 0] + os[:, 1] * d[:, 1]) + os[:, 2] * d[:, 2]
            cc = ((os[:, 0] * os[:, 0] + os[:, 1] * os[:, 1]) + os[:, 2] * os[:, 2]) - r * r
            disc = half_b * half_b - a * cc
            sq = np.sqrt(np.maximum(disc, F(0)))
            root = (-half_b - sq) / a
            out = (root < t_min) | (root > best_t)
            root2 = (-half_b + sq) / a
            root = np.where(out, root2, root)
            ok = (disc >= 0) & ~((root < t_min) | (root > best_t))
            best_t = np.where(ok, root, best_t)
            best_i = np.where(ok, np.uint32(i), best_i)
    return best_i, best_t


@pytest.mark.parametrize("scale,n,seed,builder", [(1e-2, 300, 1, "host"), (1.0, 1500, 2, "host"), (1e3, 1500, 3, "host"),
                                                  (1.0, 5000, 4, "host"), (1.0, 2000, 7, "host"), (1.0, 1500, 5, "device"),
                                                  (1e3, 5000, 6, "device")])
def test_random_sphere_fields_match_the_exhaustive_scan(scale, n, seed, builder, monkeypatch):
    monkeypatch.setenv("MIRT_BVH_BUILDER", builder)   # host: surface-area tree; device: Morton-order tree
    rng = np.random.default_rng(seed)
    centers = (rng.uniform(-10, 10, (n, 3)) * scale).astype(F)
    radii = (np.exp(rng.normal(-1.0, 1.0, n)) * scale).astype(F)          # 0.02 .. 7 times the scale: many overlaps
    centers[1] = centers[0]; radii[1] = radii[0]                           # coincident pair: the later one wins
    m = 20000
    rays = np.zeros((m, 7), F)
    rays[:, 0:3] = rng.uniform(-14, 14, (m, 3)) * scale
    rays[:, 3:6] = rng.normal(size=(m, 3)) * rng.uniform(0.05, 20.0, (m, 1))   # directions are not unit length
    rays[: m // 10, 0:3] = centers[rng.integers(0, n, m // 10)]            # a tenth start at sphere centres
    k = m // 10
    on = rng.integers(0, n, k)                                             # another tenth on sphere surfaces
    u = rng.normal(size=(k, 3)); u /= np.linalg.norm(u, axis=1, keepdims=True)
    rays[k:2 * k, 0:3] = centers[on] + (radii[on, None] * u).astype(F)
    rays[2 * k:2 * k + 50, 3:6] = [0, 0, 1]                                # a few axis-parallel ones (0 * inf in the slabs)
    ctx = Ctx()
    try:
        scene = ctx.upload(spheres=[sphere(tuple(c), float(r), order=i, ti=0, oi=i) for i, (c, r) in enumerate(zip(centers, radii))])
        ti, oi, t = ctx.trace(scene, rays)
    finally:
        ctx.close()
    want_i, want_t = scan(centers, radii, rays)
    hit = want_i != 0xFFFFFFFF
    assert 0.2 < hit.mean() < 1.0
    assert np.array_equal(np.where(ti == 0xFFFFFFFF, 0xFFFFFFFF, oi), want_i)
    assert np.array_equal(t.view(np.uint32), want_t.view(np.uint32))


@pytest.mark.parametrize("scale,seed,builder,far,n,radius", [
    (1.0, 21, "host", 30.0, 400, None), (1e3, 22, "host", 300.0, 400, None), (1e-2, 23, "device", 30.0, 400, None),
    (1.0, 24, "host", 3000.0, 400, None), (1.0, 25, "host", 3000.0, 4500, None), (10.0, 26, "device", 300.0, 4500, None),
    (1.0, 27, "host", 30000.0, 4500, None), (0.1, 28, "device", 30000.0, 4500, None),
    # every sphere of one radius: the padding of a four-child walk depends on the scene's smallest radius, and only a
    # scene without small spheres gets less of it than the radius-free bound gives
    (1.0, 29, "host", 3000.0, 4500, 0.2), (1.0, 30, "device", 300.0, 4500, 0.05), (10.0, 31, "device", 3000.0, 4500, 0.3)])
def test_rays_that_graze_spheres_match_the_exhaustive_scan(scale, seed, builder, far, n, radius, monkeypatch):
    """Rays aimed at the rim of a sphere from up to thousands of radii away, passing it within a few times the
    distance at which the f32 discriminant (objects.rs:46-50) can no longer tell a hit from a miss: the scan reports
    hits for some rays that miss the exact sphere, and the boxes of the tree have to let every one of them through.
    (far = 30000: from there the whole field lies inside the blur of every sphere in it.)"""
    monkeypatch.setenv("MIRT_BVH_BUILDER", builder)
    rng = np.random.default_rng(seed)
    # (4500 spheres: too many for the LDS copy, so the tree is folded into four-child nodes with quantised boxes and
    # walked from global memory)
    centers = (rng.uniform(-10, 10, (n, 3)) * scale).astype(F)
    radii = (np.exp(rng.normal(-2.0 if n < 1000 else -3.0, 1.2, n)) * scale).astype(F)   # 0.004 .. 5 times the scale
    if radius is not None:
        radii = np.full(n, radius * scale, F)
    m = 60000
    on = rng.integers(0, n, m)
    c, r = centers[on].astype(np.float64), radii[on].astype(np.float64)[:, None]
    u = rng.normal(size=(m, 3)); u /= np.linalg.norm(u, axis=1, keepdims=True)       # towards the rim ...
    w = rng.normal(size=(m, 3))
    v = np.cross(u, w); v /= np.linalg.norm(v, axis=1, keepdims=True)                # ... and along the tangent there
    dist = np.exp(rng.uniform(np.log(0.5), np.log(far), (m, 1))) * scale
    big = np.maximum(dist, r)
    zone = 10.0 * 2.0 ** -24 * 3.0 * big * big / r                                    # where hit and miss blur
    miss = rng.normal(size=(m, 1)) * np.minimum(2.0 * zone, 2e-3 * big)               # signed distance from the surface
    rays = np.zeros((m, 7), F)
    rays[:, 0:3] = (c + (r + miss) * u - dist * v).astype(F)
    rays[:, 3:6] = (v * rng.uniform(0.05, 20.0, (m, 1))).astype(F)
    ctx = Ctx()
    try:
        scene = ctx.upload(spheres=[sphere(tuple(cc), float(rr), order=i, ti=0, oi=i) for i, (cc, rr) in enumerate(zip(centers, radii))])
        ti, oi, t = ctx.trace(scene, rays)
    finally:
        ctx.close()
    want_i, want_t = scan(centers, radii, rays)
    grazing = want_i == on                                                            # the sphere aimed at is the one hit
    assert 0.1 < grazing.mean() < 0.9
    got_i = np.where(ti == 0xFFFFFFFF, 0xFFFFFFFF, oi)
    bad = np.nonzero(got_i != want_i)[0]
    assert bad.size == 0, (bad.size, bad[:5], got_i[bad[:5]], want_i[bad[:5]], on[bad[:5]])
    assert np.array_equal(t.view(np.uint32), want_t.view(np.uint32))


def rect_scan(axis, a0, a1, b0, b1, k, o, d, t_min, t_max):
    """XYRect / XZRect / YZRect::hit (objects.rs:196-228, 251-283, 325-357): axis 0 = xy (k on z), 1 = xz (k on y),
    2 = yz (k on x); returns (accepted, t)."""
    n, a, b = ((2, 0, 1), (1, 0, 2), (0, 1, 2))[axis]
    with np.errstate(all="ignore"):
        t = (k - o[:, n]) / d[:, n]
        pa = o[:, a] + t * d[:, a]
        pb = o[:, b] + t * d[:, b]
    bad = (t < t_min) | (t > t_max) | (pa < a0) | (pa > a1) | (pb < b0) | (pb > b1) | np.isnan(t)
    return ~bad, t


def mixed_scan(prims, rays, t_min=F(0.001)):
    """World::hit over a list of ("sphere", c, r) / ("rect", axis, a0, a1, b0, b1, k) / ("box", mn, mx) in order."""
    o, d = rays[:, 0:3].astype(F), rays[:, 3:6].astype(F)
    best_t = np.full(len(rays), np.inf, F)
    best_i = np.full(len(rays), 0xFFFFFFFF, np.uint32)
    for i, p in enumerate(prims):
        if p[0] == "sphere":
            bi, bt = scan(np.array([p[1]], F), np.array([p[2]], F), rays, t_min)   # one sphere, full range ...
            ok = (bi == 0) & ~(bt > best_t)                                          # ... then the shrinking t_max
            t = bt
        elif p[0] == "rect":
            ok, t = rect_scan(p[1], *[F(v) for v in p[2:7]], o, d, t_min, best_t)
        else:
            mn, mx = np.array(p[1], F), np.array(p[2], F)
            closest = best_t.copy()
            ok = np.zeros(len(rays), bool)
            for axis, a, b, k in ((0, 0, 1, mx[2]), (0, 0, 1, mn[2]), (1, 0, 2, mn[1]), (1, 0, 2, mx[1]),
                                  (2, 1, 2, mn[0]), (2, 1, 2, mx[0])):               # objects.rs:380-393, 396-436
                hit, tt = rect_scan(axis, mn[a], mx[a], mn[b], mx[b], k, o, d, t_min, closest)
                closest = np.where(hit, tt, closest)
                ok |= hit
            t = closest
        best_t = np.where(ok, t, best_t)
        best_i = np.where(ok, np.uint32(i), best_i)
    return best_i, best_t


@pytest.mark.parametrize("scale,seed,n,wrapped", [(1.0, 11, 900, False), (500.0, 12, 900, False), (0.05, 13, 900, False),
                                                   (1.0, 14, 6000, False), (20.0, 15, 6000, True)])
def test_random_mixed_scenes_match_the_exhaustive_scan(scale, seed, n, wrapped):
    """Spheres, rects of the three orientations and boxes in one World, rays also starting on rect planes and box faces.
    n = 6000: too many for the LDS copy, so the scene is walked from global memory through four-child nodes.
    wrapped: every other primitive sits in Translate(Rotate(.., 0 degrees), (0, 0, 0)), which changes no number
    (transforms.rs:15-24, 81-110 with sin = 0, cos = 1) but sends the ray through the wrapper chain."""
    rng = np.random.default_rng(seed)
    prims, spheres, rects, boxes = [], [], [], []
    chains = [R.Chain(0, 0), R.Chain(0, 2)] if wrapped else None
    ops = [R.ChainOp(0, (C.c_float * 3)(0, 0, 0)), R.ChainOp(1, (C.c_float * 3)(0.0, 1.0, 0))] if wrapped else ()
    spread = 10.0 if n < 1000 else 25.0
    for i in range(n):
        kind = rng.integers(0, 3)
        c = rng.uniform(-spread, spread, 3) * scale
        if kind == 0:
            r = float(np.exp(rng.normal(-0.5, 0.7)) * scale)
            prims.append(("sphere", F(c), F(r)))
            spheres.append(sphere(tuple(F(c)), float(F(r)), order=i, ti=0, oi=i, chain=int(wrapped and i % 2)))
        elif kind == 1:
            axis = int(rng.integers(0, 3))
            ext = rng.uniform(0.2, 4.0, 2) * scale
            a0, b0 = rng.uniform(-spread, spread, 2) * scale
            vals = [F(a0), F(a0 + ext[0]), F(b0), F(b0 + ext[1]), F(c[2])]
            prims.append(("rect", axis, *vals))
            rects.append(R.Rect(R.PrimHeader(0, int(wrapped and i % 2), i, 0, i, 0), axis, *[float(v) for v in vals]))
        else:
            ext = rng.uniform(0.2, 3.0, 3) * scale
            mn, mx = F(c), F(c + ext)
            prims.append(("box", mn, mx))
            boxes.append(R.Box(R.PrimHeader(0, int(wrapped and i % 2), i, 0, i, 0), (C.c_float * 3)(*mn), (C.c_float * 3)(*mx)))
    m = 12000
    rays = np.zeros((m, 7), F)
    rays[:, 0:3] = rng.uniform(-1.3 * spread, 1.3 * spread, (m, 3)) * scale
    rays[:, 3:6] = rng.normal(size=(m, 3)) * rng.uniform(0.05, 20.0, (m, 1))
    k = 0
    for p in prims:                                   # origins on faces: the next hit must not be the face left
        if p[0] == "box" and k < 2000:
            mn, mx = p[1], p[2]
            rays[k, 0:3] = [(mn[0] + mx[0]) / 2, mx[1], (mn[2] + mx[2]) / 2]          # centre of the top face
            k += 1
    ctx = Ctx()
    try:
        scene = ctx.upload(spheres=spheres, rects=rects, boxes=boxes, chains=chains, chain_ops=ops)
        ti, oi, t = ctx.trace(scene, rays)
    finally:
        ctx.close()
    want_i, want_t = mixed_scan(prims, rays)
    assert 0.3 < (want_i != 0xFFFFFFFF).mean() <= 1.0
    got_i = np.where(ti == 0xFFFFFFFF, 0xFFFFFFFF, oi)
    bad = np.nonzero(got_i != want_i)[0]
    assert bad.size == 0, (bad[:5], got_i[bad[:5]], want_i[bad[:5]], t[bad[:5]], want_t[bad[:5]])
    assert np.array_equal(t.view(np.uint32), want_t.view(np.uint32))


@pytest.mark.parametrize("scale,seed", [(1.0, 31), (300.0, 32), (0.02, 33)])
def test_rays_at_the_edges_and_corners_of_boxes_match_the_exhaustive_scan(scale, seed):
    """Box3d::hit tests its six sides one after the other (objects.rs:396-436).  Rays aimed at edges and corners, where
    two or three sides are met at (nearly) the same t, rays that start on a side or an edge, and rays that run inside
    the plane of a side (a zero direction component: 0 / 0 and x / 0 in the sides' own tests) must come out as the
    exhaustive scan has them: same box, same bits of t."""
    rng = np.random.default_rng(seed)
    prims, boxes = [], []
    for i in range(300):
        c = rng.uniform(-10, 10, 3) * scale
        ext = rng.uniform(0.2, 3.0, 3) * scale
        mn, mx = F(c), F(c + ext)
        prims.append(("box", mn, mx))
        boxes.append(R.Box(R.PrimHeader(0, 0, i, 0, i, 0), (C.c_float * 3)(*mn), (C.c_float * 3)(*mx)))
    m = 40000
    on = rng.integers(0, 300, m)
    mn = np.array([prims[i][1] for i in on], np.float64)
    mx = np.array([prims[i][2] for i in on], np.float64)
    # a point of the box with every coordinate at the minimum, at the maximum or in between: corners, edges, sides
    pick = rng.integers(0, 3, (m, 3))
    pick[np.all(pick == 2, axis=1), 0] = 0                                   # (never an inner point)
    target = np.where(pick == 0, mn, np.where(pick == 1, mx, mn + (mx - mn) * rng.uniform(0, 1, (m, 3))))
    origin = target + rng.normal(size=(m, 3)) * scale * np.exp(rng.uniform(np.log(0.1), np.log(100.0), (m, 1)))
    rays = np.zeros((m, 7), F)
    rays[:, 0:3] = origin.astype(F)
    rays[:, 3:6] = ((target - rays[:, 0:3].astype(np.float64)) * rng.uniform(0.05, 20.0, (m, 1))).astype(F)
    k = m // 8
    rays[:k, 0:3] = target[:k].astype(F)                                     # starting on the corner / edge / side itself ...
    rays[:k, 3:6] = rng.normal(size=(k, 3)).astype(F)                        # ... in any direction
    for axis in range(3):                                                    # inside the plane of a side, and along an edge
        sel = slice(k + axis * 500, k + (axis + 1) * 500)
        rays[sel, 3 + axis] = 0.0
        rays[sel, axis] = np.where(rng.integers(0, 2, 500) == 0, mn[sel, axis], mx[sel, axis]).astype(F)
    ctx = Ctx()
    try:
        scene = ctx.upload(boxes=boxes)
        ti, oi, t = ctx.trace(scene, rays)
    finally:
        ctx.close()
    want_i, want_t = mixed_scan(prims, rays)
    assert 0.3 < (want_i != 0xFFFFFFFF).mean() < 1.0
    got_i = np.where(ti == 0xFFFFFFFF, 0xFFFFFFFF, oi)
    bad = np.nonzero(got_i != want_i)[0]
    assert bad.size == 0, (bad.size, bad[:5], got_i[bad[:5]], want_i[bad[:5]], t[bad[:5]], want_t[bad[:5]])
    assert np.array_equal(t.view(np.uint32), want_t.view(np.uint32))
