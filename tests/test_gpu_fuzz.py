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
                                                  (1.0, 5000, 4, "host"), (1.0, 1500, 5, "device"),
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
