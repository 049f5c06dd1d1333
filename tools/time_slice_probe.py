#!/usr/bin/env python3
"""Moving-sphere bounds fitted to slices of the shutter interval: tree steps and primitive tests per segment and
Msamples/s for 1, 2, 4, 8 copies of the tree.  usage: time_slice_probe.py scene spp [slices ...]"""
import os
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import raytracing_amd as R

name, spp = sys.argv[1], int(sys.argv[2])
w, h = 1920, 1080
for k in sys.argv[3:] or ["1", "2", "4", "8"]:
    os.environ["MIRT_TIME_SLICES"] = k
    g = R.ExampleScene(name)
    g.render(w, h, 1, aspect=w / h)
    _, _, sc = g.render(w, h, max(1, spp // 8), aspect=w / h, flags=R.MIRT_RENDER_COUNT_WORK)
    best = None
    for _ in range(3):
        _, _, st = g.render(w, h, spp, aspect=w / h, flags=R.MIRT_RENDER_TIME_KERNELS)
        best = st["kernel_ms"][0] if best is None else min(best, st["kernel_ms"][0])
    seg = sc["segments"]
    print(f"{name}: {k} slice(s): tree steps/segment {sc['nodes_visited'] / seg:6.2f}, primitive tests/segment "
          f"{sc['prims_tested'] / seg:5.2f}, scene in LDS {st['scene_in_lds']}, {w * h * spp / best / 1e3:8.1f} Msamples/s", flush=True)
    g.close()
