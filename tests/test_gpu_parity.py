"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle and the committed fixtures.

Bars: closest-hit ids and t are compared bit for bit (f32 +,-,*,/,sqrt only); images that share the oracle's
Philox streams are compared per pixel (they differ only where libm / fused-order rounding flips a path);
images with independent seeds are compared by PSNR on the gamma-encoded RGB24 the reference produces,
tolerance 40 dB as the target metric states.
"""
import ctypes as C
from pathlib import Path

import numpy as np
import pytest

import oracle_lib as O
import raytracing_amd as R

pytestmark = pytest.mark.gpu
GOLD = Path(__file__).resolve().parent / "golden"
F = np.float32


@pytest.fixture(scope="module")
def scenes():
    cache = {}

    def get(name, seed=1):
        key = (name, seed)
        if key not in cache:
            cache[key] = (R.ExampleScene(name, scene_seed=seed), O.Scene(name, scene_seed=seed))
        return cache[key]
    yield get
    for g, o in cache.values():
        g.close()
        o.close()


def random_rays(n, lo, hi, seed, time_lo=0.0, time_hi=1.0):
    rng = np.random.default_rng(seed)
    rays = np.zeros((n, 7), F)
    rays[:, 0:3] = rng.uniform(lo, hi, (n, 3))
    d = rng.normal(size=(n, 3))
    rays[:, 3:6] = d * rng.uniform(0.2, 3.0, (n, 1))   # directions are not unit length in the reference either
    rays[:, 6] = rng.uniform(time_lo, time_hi, n)
    return rays


def skimming_rays(n, half, seed):
    """Rays that cross a sphere field of half-width `half` just above its spheres (tops at y = 0.4), sinking so
    slowly that most travel more than `half` units, many more than 1.9 * half, before they first meet one: the
    regime where |o - c|^2 - r^2 no longer resolves r^2 and the box growth is all that keeps a hit from being culled."""
    rng = np.random.default_rng(seed)
    rays = np.zeros((n, 7), F)
    angle = rng.uniform(0, 2 * np.pi, n)
    start = rng.uniform(0.95, 1.3, n) * half                # at or beyond the rim of the field
    rays[:, 0] = -np.cos(angle) * start + rng.uniform(-0.3, 0.3, n) * half
    rays[:, 2] = -np.sin(angle) * start + rng.uniform(-0.3, 0.3, n) * half
    rays[:, 1] = rng.uniform(0.45, 3.0, n)
    reach = rng.uniform(0.8, 2.6, n) * half                  # where the ray comes down to the spheres' tops
    scale = rng.uniform(0.2, 3.0, n)                         # directions are not unit length in the reference either
    rays[:, 3] = np.cos(angle) * scale
    rays[:, 5] = np.sin(angle) * scale
    rays[:, 4] = -(rays[:, 1] - 0.4) / reach * scale
    rays[:, 6] = rng.uniform(0, 1, n)
    return rays


def assert_same_hits(g, o, rays, exact_t=True):
    ti, oi, t = g.trace(rays)
    oti, ooi, ot = o.trace(rays)
    bad = np.nonzero((ti != oti) | (oi != ooi))[0]
    assert bad.size == 0, f"{bad.size} of {len(rays)} ids differ, first ray {bad[:3]}: gpu {ti[bad[:3]]},{oi[bad[:3]]} oracle {oti[bad[:3]]},{ooi[bad[:3]]}"
    if exact_t:
        assert np.array_equal(t.view(np.uint32), ot.view(np.uint32))
    else:
        assert np.allclose(t, ot, rtol=2e-5, atol=0)
    return ti, oi, t


def test_spheres_bvh_primary_hits_bit_exact_vs_fixture_and_oracle(scenes):
    """Config 2: the 600x337 pixel-centre rays of examples/spheres_bvh; ids and t bits equal the bvh.rs traversal."""
    g, o = scenes("spheres_bvh")
    cam, _ = g.camera()
    rays = R.camera_rays(cam, 600, 337)
    ocam, _ = o.camera()
    assert np.array_equal(rays, O.pixel_centre_rays(ocam, 600, 337))
    ti, oi, t = assert_same_hits(g, o, rays)
    fx = np.load(GOLD / "spheres_bvh_primary.npz")
    hit = ti != 0xFFFFFFFF
    assert np.array_equal(hit, fx["hit"])
    assert np.array_equal(ti[hit], fx["type_index"][hit].astype(np.uint32))
    assert np.array_equal(oi[hit], fx["object_index"][hit].astype(np.uint32))
    assert np.array_equal(t.view(np.uint32), fx["t_bits"])
    assert 0.5 < hit.mean() < 0.8


@pytest.mark.parametrize("name", ["spheres", "spheres_bvh", "cornel_box"])
def test_primary_and_random_rays_bit_exact(scenes, name):
    g, o = scenes(name)
    cam, _ = g.camera()
    w, h = (200, 112) if name.startswith("spheres") else (160, 160)
    assert_same_hits(g, o, R.camera_rays(cam, w, h, time=0.25))
    if name.startswith("spheres"):
        rays = np.concatenate([random_rays(60000, [-14, 0.0, -14], [14, 3, 14], 3),
                               random_rays(20000, [-300, 0.01, -300], [300, 40, 300], 4)])
    else:
        rays = random_rays(60000, [1, 1, 1], [554, 554, 554], 5)
    assert_same_hits(g, o, rays)


def test_rays_starting_on_surfaces_bit_exact(scenes):
    """Secondary-ray-like rays: origins are hit points of a first batch (grazing and self-intersection cases)."""
    g, o = scenes("spheres")
    first = random_rays(50000, [-12, 0.05, -12], [12, 4, 12], 8)
    _, _, t = g.trace(first)
    ok = np.isfinite(t)
    rays = first[ok].copy()
    rays[:, 0:3] = (first[ok, 0:3] + first[ok, 3:6] * t[ok, None]).astype(F)
    rays[:, 3:6] = np.random.default_rng(9).normal(size=(ok.sum(), 3))
    assert_same_hits(g, o, rays)


def test_final_scene_hits(scenes):
    """Instanced spheres, 400 boxes, rects and the two constant media.  Media t depends on log2 (libm vs
    device), so t is compared to 2e-5 relative there and bit for bit on surfaces."""
    g, o = scenes("final_scene")
    cam, _ = g.camera()
    rays = np.concatenate([R.camera_rays(cam, 200, 200), random_rays(40000, [-200, 20, -200], [600, 540, 600], 6)])
    ti, oi, t = g.trace(rays)
    oti, ooi, ot = o.trace(rays)
    same = (ti == oti) & (oi == ooi)
    # a medium boundary decision can flip on the last bit of log2: allow a handful, none on surfaces
    medium = (oti == 5) | (ti == 5)          # ConstantMedium is the sixth concrete type added (final_scene.rs:122-140)
    assert np.all(same | medium)
    assert np.mean(~same) < 2e-4
    surf = same & ~medium
    assert np.array_equal(t[surf].view(np.uint32), ot[surf].view(np.uint32))
    assert np.allclose(t[same], ot[same], rtol=2e-5)
    assert medium.mean() > 0.02      # the fog is actually exercised


def test_synthetic_bvh_scene_bit_exact(scenes):
    """The random-spheres generator widened to 81x81 cells (6.5 k spheres) through BVH::from_world."""
    g, o = scenes("synthetic_bvh:40")
    assert g.primitive_count() == o.object_count() > 6000
    cam, _ = g.camera()
    rays = np.concatenate([R.camera_rays(cam, 160, 90), random_rays(30000, [-42, 0.0, -42], [42, 3, 42], 12)])
    assert_same_hits(g, o, rays)


@pytest.mark.parametrize("name,size,spp", [("spheres", (150, 84), 16), ("spheres_bvh", (90, 50), 8),
                                           ("cornel_box", (96, 96), 32), ("final_scene", (96, 96), 32),
                                           ("synthetic_bvh:40", (96, 54), 4)])  # too big for LDS: tables in global memory
def test_same_seed_images_match_per_pixel(scenes, name, size, spp):
    """The GPU and the oracle read the same Philox streams, so the images agree path by path."""
    g, o = scenes(name)
    w, h = size
    aspect = 0.0 if name.startswith(("spheres", "synthetic")) else w / h
    rgb, sums, st = g.render(w, h, spp, aspect=aspect, seed=21, flags=R.MIRT_RENDER_COUNT_WORK)
    orgb, osums, oseg = o.render(w, h, spp, aspect=aspect, seed=21)
    assert abs(st["segments"] - oseg) <= max(8, oseg // 20000)
    diff = np.abs(rgb.astype(int) - orgb.astype(int))
    psnr = float(O.psnr(rgb, orgb))
    # a path whose medium/texture decision flips on the last bit of log2/sin (device libm vs glibc) ends
    # elsewhere; in the dark final scene one such path moves a pixel visibly, hence the lower bar there
    assert psnr > (45.0 if name == "final_scene" else 55.0), psnr
    assert float(np.mean(diff > 1)) < 5e-3
    ok = np.isfinite(osums) & np.isfinite(sums)
    assert ok.mean() > 0.999
    rel = np.abs(sums[ok] - osums[ok]) / (np.abs(osums[ok]) + 1e-3)
    assert np.quantile(rel, 0.99) < 1e-3


@pytest.mark.parametrize("size,spp", [((37, 23), 5), ((2, 2), 64), ((130, 7), 3), ((8, 61), 2)])
def test_ragged_frames_match_the_oracle_per_pixel(scenes, size, spp):
    """Frames whose sides are no multiple of the 8x8 tiles paths are started in, down to the smallest frame the
    renderer accepts, with the smallest pool and a pool left to the library: same seed, same image as the oracle."""
    g, o = scenes("spheres")
    w, h = size
    orgb, osums, _ = o.render(w, h, spp, seed=5)
    for pool in (0, 1):                                    # (1: rounded up to the smallest pool there is)
        rgb, sums, st = g.render(w, h, spp, seed=5, path_pool=pool)
        assert st["samples"] == w * h * spp
        ok = np.isfinite(osums) & np.isfinite(sums)
        assert ok.mean() > 0.99
        rel = np.abs(sums[ok] - osums[ok]) / (np.abs(osums[ok]) + 1e-3)
        assert np.quantile(rel, 0.98) < 1e-3
        assert float(np.mean(np.abs(rgb.astype(int) - orgb.astype(int)) > 1)) < 0.02


@pytest.mark.parametrize("name", ["spheres", "cornel_box", "final_scene"])
def test_converged_image_psnr_vs_cpu_fixture(scenes, name):
    """Independent seeds: GPU render (4x the fixture's spp) against the committed converged CPU image,
    PSNR >= 40 dB on RGB24.  Pixels that drew a NaN sample in either render are black by the reference's
    own rule (renderer.rs:241-252) at random positions, so they are excluded (and must stay rare)."""
    fx = np.load(GOLD / f"{name}_converged.npz")
    w, h, spp = int(fx["width"]), int(fx["height"]), int(fx["spp"])
    g, _ = scenes(name)
    assert int(fx["seed"]) != 99
    rgb, sums, _ = g.render(w, h, 4 * spp, depth=int(fx["depth"]), aspect=float(fx["aspect"]), seed=99)
    poisoned = fx["nan_mask"] | ~np.isfinite(sums).all(axis=2)
    assert poisoned.mean() < 0.01
    assert np.all(rgb[~np.isfinite(sums).all(axis=2)] == 0)
    psnr = float(O.psnr(rgb[~poisoned], fx["rgb"][~poisoned]))
    assert psnr >= 40.0, psnr


def test_sample_ranges_add_up_and_runs_repeat(scenes):
    """Sharding property at the benchmark's resolution: samples [0,4) + [4,6) == [0,6) up to f32 summation
    order, and two identical calls agree (atomics reorder the per-pixel sums only)."""
    g, _ = scenes("final_scene")
    w, h = 1920, 1080
    full = g.render(w, h, 6, aspect=w / h, seed=3)[1]
    again = g.render(w, h, 6, aspect=w / h, seed=3, path_pool=1 << 20)[1]
    lo = g.render(w, h, 4, aspect=w / h, seed=3, first_sample=0)[1]
    hi = g.render(w, h, 2, aspect=w / h, seed=3, first_sample=4)[1]
    ok = np.isfinite(full)
    assert ok.mean() > 0.9999
    assert np.array_equal(np.isfinite(again), ok)
    assert np.allclose(again[ok], full[ok], rtol=1e-4, atol=1e-5)
    both = ok & np.isfinite(lo) & np.isfinite(hi)
    assert np.allclose((lo + hi)[both], full[both], rtol=1e-4, atol=1e-5)
    other = g.render(w, h, 6, aspect=w / h, seed=4)[1]
    assert not np.allclose(other[ok], full[ok], rtol=1e-4, atol=1e-5)


def test_resolve_matches_reference_encoding(scenes):
    """renderer.rs:232-259 on the device against the oracle's encoder, special values included."""
    import torch
    g, _ = scenes("cornel_box")
    rng = np.random.default_rng(2)
    w, h, spp = 64, 48, 7
    sums = (rng.uniform(0, 12, (h, w, 3)) ** 2).astype(F)
    sums.reshape(-1)[:8] = [0.0, -0.0, np.nan, np.inf, -np.inf, -3.0, 7.0 * 0.999 ** 2, 1e-30]
    d_sum = torch.from_numpy(sums).cuda()
    d_rgb = torch.zeros((h, w, 3), dtype=torch.uint8, device="cuda")
    # (no stream = the null stream, torch's own: the call runs behind the copy and the fill that made the buffers)
    g.resolve(d_sum.data_ptr(), w, h, spp, d_rgb.data_ptr())
    torch.cuda.synchronize()
    assert np.array_equal(d_rgb.cpu().numpy(), O.encode(sums, spp))


def test_depth_limits_and_background(scenes):
    g, o = scenes("spheres")
    rgb, sums, st = g.render(32, 18, 4, depth=0)
    assert not sums.any() and not rgb.any()
    rgb, sums, st = g.render(32, 18, 4, depth=1, seed=5)
    orgb, osums, _ = o.render(32, 18, 4, depth=1, seed=5)
    assert np.allclose(sums, osums, rtol=1e-5, atol=1e-6)
    assert np.allclose(sums[0] / 4, 0.8, atol=1e-6)     # sky row: background (renderer.rs:187-189)


@pytest.mark.parametrize("name,lo,hi", [("spheres_bvh", [-12, 0.0, -12], [12, 3, 12]),
                                        ("cornel_box", [0, 0, -800], [555, 555, 555]),
                                        ("synthetic_bvh:40", [-42, 0.0, -42], [42, 3, 42])])
def test_device_built_tree_finds_the_same_hits(name, lo, hi, monkeypatch):
    """The closest hit does not depend on the tree: with the BVH built on the GPU (Morton order, balanced levels)
    ids and t are still the oracle's, bit for bit, and the same-seed image is the one the host-built tree gives."""
    monkeypatch.setenv("MIRT_BVH_BUILDER", "device")
    g, o = R.ExampleScene(name), O.Scene(name)
    monkeypatch.delenv("MIRT_BVH_BUILDER")
    try:
        cam, _ = g.camera()
        rays = np.concatenate([R.camera_rays(cam, 120, 68), random_rays(20000, lo, hi, 31)])
        assert_same_hits(g, o, rays)
        aspect = 1.0 if name == "cornel_box" else 0.0
        rgb, sums, _ = g.render(96, 96, 4, aspect=aspect, seed=9)
        h = R.ExampleScene(name)          # host-built tree
        rgb2, sums2, _ = h.render(96, 96, 4, aspect=aspect, seed=9)
        h.close()
        ok = np.isfinite(sums) & np.isfinite(sums2)
        assert ok.mean() > 0.995
        assert np.allclose(sums[ok], sums2[ok], rtol=1e-4, atol=1e-5)
    finally:
        g.close()
        o.close()


@pytest.mark.parametrize("budget", ["3", "40", "1000"])
def test_walks_put_aside_and_resumed_find_the_same_hits(budget, monkeypatch):
    """Large scenes give a ray a budget of steps per round; a walk that uses it up is put aside (place in the tree,
    stack, closest hit so far) and goes on in the next round (forced here on the 6.5 k-sphere scene, with budgets
    from "nearly every ray, many times" to "hardly any").  The steps are the same steps in the same order, so
    the image is the one whole-ray walks give, and the oracle's."""
    name, w, h, spp = "synthetic_bvh:40", 96, 54, 4
    g, o = R.ExampleScene(name), O.Scene(name)
    try:
        monkeypatch.setenv("MIRT_STEP_BUDGET", "0")
        whole, whole_sums, st0 = g.render(w, h, spp, seed=21, flags=R.MIRT_RENDER_COUNT_WORK)
        monkeypatch.setenv("MIRT_STEP_BUDGET", budget)
        rgb, sums, st = g.render(w, h, spp, seed=21, flags=R.MIRT_RENDER_COUNT_WORK)
        orgb, osums, oseg = o.render(w, h, spp, seed=21)
    finally:
        g.close()
        o.close()
    assert st["segments"] == st0["segments"]
    # not one step more or less than whole-ray walks take
    assert st["nodes_visited"] == st0["nodes_visited"] and st["prims_tested"] == st0["prims_tested"]
    if budget != "1000":
        assert st["iterations"] > st0["iterations"]          # more rounds: rays were carried over
    ok = np.isfinite(sums) & np.isfinite(whole_sums)
    assert np.array_equal(np.isfinite(sums), np.isfinite(whole_sums))
    assert np.allclose(sums[ok], whole_sums[ok], rtol=1e-5, atol=1e-6)
    assert float(O.psnr(rgb, orgb)) > 55.0
    assert abs(st["segments"] - oseg) <= max(8, oseg // 20000)


def test_a_full_save_area_loses_no_ray(monkeypatch):
    """More walks use up their budget in a round than the region's save area holds (a quarter of the region): the
    ones that find no room walk on instead, and every path is still traced -- same segments, same steps, same image
    as with whole-ray walks."""
    name, w, h, spp = "synthetic_bvh:40", 640, 360, 4
    g = R.ExampleScene(name)
    try:
        monkeypatch.setenv("MIRT_STEP_BUDGET", "0")
        _, whole_sums, st0 = g.render(w, h, spp, seed=9, path_pool=1 << 20, flags=R.MIRT_RENDER_COUNT_WORK)
        monkeypatch.setenv("MIRT_STEP_BUDGET", "2")        # every walk is spent at once: four times the save area
        _, sums, st = g.render(w, h, spp, seed=9, path_pool=1 << 20, flags=R.MIRT_RENDER_COUNT_WORK)
    finally:
        g.close()
    assert st["path_pool"] == 1 << 20                      # regions of 256 slots, save areas of 64
    assert st["segments"] == st0["segments"]
    assert st["nodes_visited"] == st0["nodes_visited"] and st["prims_tested"] == st0["prims_tested"]
    ok = np.isfinite(sums) & np.isfinite(whole_sums)
    assert np.array_equal(np.isfinite(sums), np.isfinite(whole_sums))
    assert np.allclose(sums[ok], whole_sums[ok], rtol=1e-5, atol=1e-6)


def test_large_scene_library_defaults(scenes):
    """A scene of more than 2^18 primitives with nothing overridden: the library picks the device (Morton order)
    builder and gives camera paths a step budget per round.  Closest-hit ids and t bits of caller rays against the oracle's
    bvh.rs tournament, then a same-seed render whose round count shows rays were carried from round to round."""
    import os
    for var in ("MIRT_BVH_BUILDER", "MIRT_STEP_BUDGET", "MIRT_LEAF_MAX", "MIRT_REFILL_MIN", "MIRT_NODE_MIN"):
        assert var not in os.environ
    g, o = scenes("synthetic_bvh:256")                      # 513 x 513 cells
    n = g.primitive_count()
    assert n == o.object_count() and n >= 1 << 18
    cam, _ = g.camera()
    rays = np.concatenate([R.camera_rays(cam, 64, 36),
                           random_rays(1200, [-250, 0.0, -250], [250, 3, 250], 31),      # inside the field
                           random_rays(500, [-400, 0.01, -400], [400, 60, 400], 32)])     # above and around it
    assert_same_hits(g, o, rays)
    depth = 6
    # every path of this frame is in flight at once, so whole-ray walks would take one round per bounce
    _, _, st = g.render(480, 270, 1, depth=depth, seed=21)
    assert not st["scene_in_lds"]
    assert st["iterations"] > depth, "no ray was carried over to another round"
    rgb, sums, st = g.render(32, 18, 1, depth=depth, seed=21)
    orgb, osums, _ = o.render(32, 18, 1, depth=depth, seed=21)
    ok = np.isfinite(osums).all(axis=2)
    close = np.isclose(sums, osums, rtol=1e-4, atol=1e-5).all(axis=2)
    assert np.mean(close[ok]) > 0.98                       # a libm rounding may flip a path


def test_million_sphere_field_against_the_linear_scan():
    """A field of 1 002 003 spheres (the synthetic scene at half = 500) on the library's defaults, ids and t bits
    against the oracle's linear World::hit scan (world.rs:75-94: every sphere against every ray): camera rays,
    rays inside the field, and rays that skim the field for up to 1300 units before they meet a sphere."""
    g, o = R.ExampleScene("synthetic:500"), O.Scene("synthetic:500")
    try:
        assert g.primitive_count() == o.object_count() == 1002003
        cam, _ = g.camera()
        rays = np.concatenate([R.camera_rays(cam, 32, 18), skimming_rays(500, 500, 7),
                               random_rays(300, [-500, 0.0, -500], [500, 2, 500], 8)])
        ti, oi, t = assert_same_hits(g, o, rays)
        hit = ti != 0xFFFFFFFF
        travelled = np.linalg.norm(rays[:, 3:6], axis=1) * np.where(hit, t, 0.0)
        assert (travelled > 500).sum() > 100 and travelled.max() > 900, (int((travelled > 500).sum()), float(travelled.max()))
    finally:
        g.close()
        o.close()


@pytest.mark.parametrize("slices", ["4", "7"])
def test_tree_copies_fitted_to_time_slices_find_the_same_hits(slices, monkeypatch):
    """MIRT_TIME_SLICES keeps several copies of the tree, each fitted to where the moving spheres are during a part of
    the shutter interval (the other reading of the time0 / time1 arguments BVH::from_world ignores, bvh.rs:24-27);
    a ray walks the copy of its time.  Same closest hits, bit for bit, at every ray time including both ends."""
    monkeypatch.setenv("MIRT_TIME_SLICES", slices)
    g, o = R.ExampleScene("spheres_bvh"), O.Scene("spheres_bvh")
    try:
        cam, _ = g.camera()
        rays = np.concatenate([R.camera_rays(cam, 120, 68, time=0.0), R.camera_rays(cam, 120, 68, time=1.0),
                               R.camera_rays(cam, 120, 68, time=0.25), random_rays(40000, [-14, 0.0, -14], [14, 3, 14], 41)])
        assert_same_hits(g, o, rays)
    finally:
        g.close()
        o.close()
