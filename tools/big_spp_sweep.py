#!/usr/bin/env python3
"""A large synthetic scene built once, then 1080p frames of growing sample counts: what of a frame's time is a fixed
cost (first fill, drain, the longest path's own chain of rounds) and what scales with the samples.
usage: big_spp_sweep.py half spp [spp ...]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import raytracing_amd as R

half = int(sys.argv[1])
g = R.ExampleScene(f"synthetic_bvh:{half}")
print(f"synthetic_bvh:{half}: {g.primitive_count()} primitives, library {R.LIBMIRT}", flush=True)
g.render(1920, 1080, 1)
for spp in (int(v) for v in sys.argv[2:]):
    best, rounds = None, 0
    for _ in range(2):
        _, _, st = g.render(1920, 1080, spp, flags=R.MIRT_RENDER_TIME_KERNELS)
        ms = st["kernel_ms"][0]
        if best is None or ms < best:
            best, rounds = ms, st["iterations"]
    print(f"  {spp:4d} spp: {best:8.2f} ms, {1920 * 1080 * spp / best / 1e3:8.1f} Msamples/s, {rounds:4d} rounds, pool {st['path_pool']}", flush=True)
