#!/usr/bin/env python3
"""Many short renders with random frame sizes, sample counts, depths, pools and estimator flags on every example scene
(and one scene walked from global memory): each must return, count every sample and leave no value of the wrong sign.
usage: stress_render.py [runs] [seed]"""
import sys
import time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import raytracing_amd as R

runs = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
names = ["spheres", "spheres_bvh", "cornel_box", "final_scene", "synthetic_bvh:40"]
scenes = {n: R.ExampleScene(n) for n in names}
t0 = time.time()
for i in range(runs):
    name = names[int(rng.integers(0, len(names)))]
    g = scenes[name]
    w, h = int(rng.integers(2, 260)), int(rng.integers(2, 200))
    spp = int(rng.integers(1, 24))
    depth = int(rng.integers(0, 14))
    pool = int(rng.choice([0, 1, 262144, 524288, 1 << 20, 3 << 20]))
    flags = 0
    if rng.random() < 0.3:
        flags |= R.MIRT_RENDER_COUNT_WORK
    seed = int(rng.integers(0, 1 << 31))
    first = int(rng.integers(0, 1000))
    rgb, sums, st = g.render(w, h, spp, depth=depth, seed=seed, first_sample=first, path_pool=pool, flags=flags)
    assert st["samples"] == w * h * spp, (name, w, h, spp, st)
    finite = np.isfinite(sums)
    assert np.all(sums[finite] >= 0.0), (name, w, h, spp, depth)
    if depth == 0:
        assert not sums.any()
    if flags & R.MIRT_RENDER_COUNT_WORK and depth > 0:
        assert st["segments"] >= w * h * spp, (name, st)
    if i % 20 == 19:
        print(f"{i + 1} renders, {time.time() - t0:.1f} s, last: {name} {w}x{h} {spp} spp depth {depth} pool {st['path_pool']} "
              f"{st['iterations']} rounds", flush=True)
print("stress ok:", runs, "renders")
