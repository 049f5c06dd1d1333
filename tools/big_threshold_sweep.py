#!/usr/bin/env python3
"""A large synthetic scene built once, then a 1080p frame for each pair of the walk's thresholds.
usage: big_threshold_sweep.py half spp refill:node [refill:node ...]   (MIRT_REFILL_MIN, MIRT_NODE_MIN, read per render)"""
import os
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import raytracing_amd as R

half, spp = int(sys.argv[1]), int(sys.argv[2])
g = R.ExampleScene(f"synthetic_bvh:{half}")
print(f"synthetic_bvh:{half}: {g.primitive_count()} primitives, library {R.LIBMIRT}", flush=True)
for pair in sys.argv[3:]:
    refill, node = pair.split(":")
    os.environ["MIRT_REFILL_MIN"], os.environ["MIRT_NODE_MIN"] = refill, node
    g.render(1920, 1080, 1)
    best = None
    for _ in range(2):
        _, _, st = g.render(1920, 1080, spp, flags=R.MIRT_RENDER_TIME_KERNELS)
        ms = st["kernel_ms"][0]
        best = ms if best is None else min(best, ms)
    _, _, sc = g.render(1920, 1080, spp, flags=R.MIRT_RENDER_COUNT_WORK)
    cyc = sc["phase_cycles"]
    share = [c / max(1, cyc[5]) for c in cyc[:5]]
    print(f"  refill {refill:>2} node {node:>2}: {1920 * 1080 * spp / best / 1e3:8.1f} Msamples/s, {best:8.2f} ms, lanes at work "
          f"{sc['nodes_visited'] / max(1, 64 * sc['node_wave_steps']):.2f} / {sc['prims_tested'] / max(1, 64 * sc['prim_wave_steps']):.2f}, "
          f"wave time: fetch {share[0]:.2f} tree {share[1]:.2f} tests {share[2]:.2f} shade {share[3]:.2f} hand-in {share[4]:.2f}", flush=True)
