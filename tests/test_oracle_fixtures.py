"""The committed fixtures are reproducible outputs of the oracle (tools/make_goldens.py made them)."""
from pathlib import Path

import numpy as np
import pytest

import oracle_lib as O

GOLD = Path(__file__).resolve().parent / "golden"


def test_primary_hit_fixture_is_what_the_oracle_traces():
    fx = np.load(GOLD / "spheres_bvh_primary.npz")
    s = O.Scene("spheres_bvh", scene_seed=int(fx["scene_seed"]))
    cam, _ = s.camera()
    rays = O.pixel_centre_rays(cam, int(fx["width"]), int(fx["height"]))
    ti, oi, t = s.trace(rays)
    hit = ti != 0xFFFFFFFF
    assert np.array_equal(hit, fx["hit"])
    assert np.array_equal(ti[hit], fx["type_index"][hit].astype(np.uint32))
    assert np.array_equal(oi[hit], fx["object_index"][hit].astype(np.uint32))
    assert np.array_equal(t.view(np.uint32), fx["t_bits"])


@pytest.mark.parametrize("name,spp,floor", [("spheres", 64, 24.0), ("cornel_box", 256, 18.0), ("final_scene", 64, 14.0)])
def test_converged_fixture_is_the_limit_of_low_spp_renders(name, spp, floor):
    """A cheap render with another seed is a noisy version of the converged fixture, and gets closer with more spp."""
    fx = np.load(GOLD / f"{name}_converged.npz")
    w, h = int(fx["width"]), int(fx["height"])
    s = O.Scene(name, scene_seed=int(fx["scene_seed"]))
    lo = s.render(w, h, max(1, spp // 8), depth=int(fx["depth"]), aspect=float(fx["aspect"]), seed=5)[0]
    hi = s.render(w, h, spp, depth=int(fx["depth"]), aspect=float(fx["aspect"]), seed=5)[0]
    p_lo, p_hi = O.psnr(lo, fx["rgb"]), O.psnr(hi, fx["rgb"])
    assert p_hi > floor and p_hi > p_lo + 2.0
