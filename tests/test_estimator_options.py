"""The opt-in estimator changes (even light sampling, dropped non-finite samples, stratified pixel jitter).

None of them exists in the reference, so there is nothing of the reference's to pin them on: the CPU tests check
the oracle's restatement of each rule against what the rule promises (same limit image, fewer black pixels, less
noise), the GPU test checks that the device library follows the oracle sample for sample when a rule is on."""
from pathlib import Path

import numpy as np
import pytest

import oracle_lib as O

GOLD = Path(__file__).resolve().parent / "golden"
EVEN_LIGHTS, DROP_NONFINITE = 4, 8
POISONED_FIRST = 808  # cornel_box 64x64, seed 21: samples [808, 816) hold a NaN sample (found by search)


def strata(n):
    return n << 8


def finite_rows(*images):
    ok = np.ones(images[0].shape[:2], bool)
    for im in images:
        ok &= np.isfinite(im).all(axis=2)
    return ok


def test_flag_values_match_the_header():
    import raytracing_amd as R
    text = (Path(R.ROOT) / "include" / "mirt.h").read_text()
    assert "MIRT_RENDER_EVEN_LIGHTS = 4u" in text and "MIRT_RENDER_DROP_NONFINITE = 8u" in text
    assert "MIRT_RENDER_STRATA_SHIFT = 8" in text
    assert (R.MIRT_RENDER_EVEN_LIGHTS, R.MIRT_RENDER_DROP_NONFINITE, R.MIRT_RENDER_STRATA(3)) == (4, 8, 0x300)
    with pytest.raises(ValueError):
        R.MIRT_RENDER_STRATA(256)


def test_one_stratum_is_the_reference_jitter():
    s = O.Scene("spheres")
    a = s.render(60, 34, 4, seed=2)[1]
    b = s.render(60, 34, 4, seed=2, estimator=strata(1))[1]
    assert np.array_equal(a, b)


def test_stratified_jitter_has_the_same_limit_and_less_noise():
    """Against the converged fixture: 2x2 strata at 4 spp are at least as close as plain jitter at 4 spp, and a
    sample range [4, 8) covers the cells as [0, 4) does."""
    fx = np.load(GOLD / "spheres_converged.npz")
    w, h = int(fx["width"]), int(fx["height"])
    s = O.Scene("spheres", scene_seed=int(fx["scene_seed"]))
    kw = dict(depth=int(fx["depth"]), aspect=float(fx["aspect"]))
    ref = fx["rgb"].astype(np.float64)
    err_plain, err_strat = [], []
    for seed in (11, 12, 13):
        plain = s.render(w, h, 4, seed=seed, **kw)[0]
        strat = s.render(w, h, 4, seed=seed, estimator=strata(2), **kw)[0]
        err_plain.append(np.mean((plain - ref) ** 2))
        err_strat.append(np.mean((strat - ref) ** 2))
    assert np.mean(err_strat) < np.mean(err_plain)
    first = s.render(w, h, 4, seed=11, estimator=strata(2), **kw)[1]
    later = s.render(w, h, 4, seed=11, first_sample=4, estimator=strata(2), **kw)[1]
    assert not np.array_equal(first, later)
    assert abs(first.mean() - later.mean()) < 0.02 * first.mean()


def test_dropping_non_finite_samples_only_touches_poisoned_pixels():
    """cornel_box: a path that starts inside the glass sphere gets a NaN pdf from the sphere light
    (objects.rs:82-94) and blackens its pixel; with the rule on the pixel keeps its other samples."""
    s = O.Scene("cornel_box")
    kw = dict(aspect=1.0, seed=21, first_sample=POISONED_FIRST)   # sample 812 of one pixel is such a path
    plain = s.render(64, 64, 8, **kw)[1]
    kept = s.render(64, 64, 8, estimator=DROP_NONFINITE, **kw)[1]
    poisoned = ~np.isfinite(plain).all(axis=2)
    assert poisoned.any()
    assert np.isfinite(kept).all()
    assert np.array_equal(plain[~poisoned], kept[~poisoned])
    assert (kept[poisoned] >= 0).all()


def test_even_lights_has_the_same_limit_in_cornel_box():
    """cornel_box's light list holds one object of each of two types, so picking the type first or picking among
    all objects is the same distribution: the images differ sample by sample (one draw instead of two) and agree
    in the mean."""
    s = O.Scene("cornel_box")
    kw = dict(aspect=1.0, seed=9)
    a = s.render(64, 64, 64, **kw)[1]
    b = s.render(64, 64, 64, estimator=EVEN_LIGHTS, **kw)[1]
    ok = finite_rows(a, b)
    assert ok.mean() > 0.99
    assert not np.array_equal(a[ok], b[ok])
    assert abs(a[ok].mean() - b[ok].mean()) < 0.03 * a[ok].mean()


@pytest.mark.gpu
@pytest.mark.parametrize("name,size,spp,first,flags", [
    ("cornel_box", (96, 96), 32, 0, EVEN_LIGHTS),
    ("cornel_box", (64, 64), 8, POISONED_FIRST, DROP_NONFINITE),
    ("cornel_box", (64, 64), 18, 0, strata(3)),
    ("final_scene", (96, 96), 16, 0, EVEN_LIGHTS | DROP_NONFINITE | strata(4)),
    ("spheres", (120, 68), 8, 0, strata(2)),
])
def test_device_follows_the_oracle_sample_for_sample(name, size, spp, first, flags):
    """Same Philox streams, same rules: the GPU image is the oracle's, pixel by pixel, with each rule on."""
    import raytracing_amd as R
    g, o = R.ExampleScene(name), O.Scene(name)
    try:
        w, h = size
        aspect = 0.0 if name == "spheres" else w / h
        rgb, sums, st = g.render(w, h, spp, aspect=aspect, seed=21, first_sample=first,
                                 flags=flags | R.MIRT_RENDER_COUNT_WORK)
        orgb, osums, oseg = o.render(w, h, spp, aspect=aspect, seed=21, first_sample=first, estimator=flags)
        plain = o.render(w, h, spp, aspect=aspect, seed=21, first_sample=first)[1]
    finally:
        g.close()
        o.close()
    assert not np.array_equal(plain, osums, equal_nan=True)   # the rule changed something
    assert abs(st["segments"] - oseg) <= max(8, oseg // 20000)
    assert np.array_equal(np.isfinite(sums).all(axis=2), np.isfinite(osums).all(axis=2)) or \
        np.mean(np.isfinite(sums).all(axis=2) != np.isfinite(osums).all(axis=2)) < 1e-3
    if flags & DROP_NONFINITE:
        assert np.isfinite(sums).all()
    diff = np.abs(rgb.astype(int) - orgb.astype(int))
    assert float(O.psnr(rgb, orgb)) > (45.0 if name == "final_scene" else 55.0)
    assert float(np.mean(diff > 1)) < 5e-3
    ok = np.isfinite(osums) & np.isfinite(sums)
    rel = np.abs(sums[ok] - osums[ok]) / (np.abs(osums[ok]) + 1e-3)
    assert np.quantile(rel, 0.99) < 1e-3
