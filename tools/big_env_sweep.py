#!/usr/bin/env python3
"""A large synthetic scene built once, then a 1080p frame for each value of one of the library's tuning switches.
usage: big_env_sweep.py half spp VAR value [value ...]     VAR = an environment switch read per render
       (MIRT_STEP_BUDGET, MIRT_DRAIN_MIN, MIRT_REFILL_MIN, MIRT_NODE_MIN) or POOL (paths in flight)"""
import os
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import raytracing_amd as R

half, spp, var = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
g = R.ExampleScene(f"synthetic_bvh:{half}")
print(f"synthetic_bvh:{half}: {g.primitive_count()} primitives, library {R.LIBMIRT}", flush=True)
for value in sys.argv[4:]:
    pool = 0
    if var == "POOL":
        pool = int(value)
    else:
        os.environ[var] = value
    g.render(1920, 1080, 1, path_pool=pool)
    best = None
    for _ in range(2):
        _, _, st = g.render(1920, 1080, spp, flags=R.MIRT_RENDER_TIME_KERNELS, path_pool=pool)
        ms = st["kernel_ms"][0]
        best = ms if best is None else min(best, ms)
    _, _, sc = g.render(1920, 1080, spp, flags=R.MIRT_RENDER_COUNT_WORK, path_pool=pool)
    cyc = sc["phase_cycles"]
    seg = max(1, sc["segments"])
    share = [c / max(1, cyc[5]) for c in cyc[:5]]
    print(f"  {var}={value:>9}: {1920 * 1080 * spp / best / 1e3:8.1f} Msamples/s, {best:8.2f} ms, {st['iterations']:4d} rounds, "
          f"steps/segment {sc['nodes_visited'] / seg:.1f} + {sc['prims_tested'] / seg:.1f}, lanes at work "
          f"{sc['nodes_visited'] / max(1, 64 * sc['node_wave_steps']):.2f} / {sc['prims_tested'] / max(1, 64 * sc['prim_wave_steps']):.2f}, "
          f"wave time: fetch {share[0]:.2f} tree {share[1]:.2f} tests {share[2]:.2f} shade {share[3]:.2f} hand-in {share[4]:.2f}",
          flush=True)
