#!/usr/bin/env python3
"""Full-size frames for a few minutes: every example scene and two scenes walked from global memory, random sample
counts and pools, each frame checked for its sample count and (where a fixture of that frame exists) its image."""
import sys
import time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import raytracing_amd as R

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(23)
names = ["spheres", "spheres_bvh", "cornel_box", "final_scene", "synthetic_bvh:40", "synthetic_bvh:150"]
scenes = {n: R.ExampleScene(n) for n in names}
t0 = time.time()
frames = 0
samples = 0
while time.time() - t0 < seconds:
    name = names[int(rng.integers(0, len(names)))]
    g = scenes[name]
    w, h = (1000, 1000) if name == "cornel_box" else (1920, 1080)
    spp = int(rng.integers(1, 96))
    pool = int(rng.choice([0, 1 << 20, 1 << 22, 1 << 23, 1 << 25]))
    _, sums, st = g.render(w, h, spp, aspect=w / h if name in ("cornel_box", "final_scene") else 0.0,
                           seed=int(rng.integers(0, 1 << 30)), path_pool=pool)
    assert st["samples"] == w * h * spp
    assert np.isfinite(sums).mean() > 0.999 and float(sums[np.isfinite(sums)].min()) >= 0.0
    frames += 1
    samples += w * h * spp
    if frames % 10 == 0:
        print(f"{frames} frames, {time.time() - t0:.0f} s, {samples / 1e9:.1f} G samples, last {name} {spp} spp pool {st['path_pool']} "
              f"{st['render_ms']:.1f} ms", flush=True)
print("soak ok:", frames, "frames")
