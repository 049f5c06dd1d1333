#!/usr/bin/env python3
"""Step budgets on a large synthetic scene: Msamples/s of a 1080p frame for each budget, scene built once.
usage: big_budget_sweep.py half spp budget [budget ...]   (budget 0 = whole-ray walks)"""
import os
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import raytracing_amd as R

half, spp = int(sys.argv[1]), int(sys.argv[2])
g = R.ExampleScene(f"synthetic_bvh:{half}")
print(f"synthetic_bvh:{half}: {g.primitive_count()} primitives, library {R.LIBMIRT}", flush=True)
for budget in sys.argv[3:]:
    os.environ["MIRT_STEP_BUDGET"] = budget
    g.render(1920, 1080, 1)
    best = None
    for _ in range(2):
        _, _, st = g.render(1920, 1080, spp, flags=R.MIRT_RENDER_TIME_KERNELS)
        ms = st["kernel_ms"][0]
        best = ms if best is None else min(best, ms)
    _, _, sc = g.render(1920, 1080, spp, flags=R.MIRT_RENDER_COUNT_WORK)
    cyc = sc["phase_cycles"]
    busy = sum(cyc[:5]) / max(1, cyc[5])
    seg = max(1, sc["segments"])
    share = [c / max(1, cyc[5]) for c in cyc[:5]]
    print(f"  budget {budget:>5}: steps/segment {sc['nodes_visited'] / seg:.1f} + {sc['prims_tested'] / seg:.1f} tests; wave time: "
          f"start+fetch {share[0]:.2f}, tree {share[1]:.2f}, tests {share[2]:.2f}, shade {share[3]:.2f}, hand-in {share[4]:.2f}")
    print(f"  budget {budget:>5}: {1920 * 1080 * spp / best / 1e3:8.1f} Msamples/s, {best:8.2f} ms, {st['iterations']:4d} rounds, "
          f"waves busy {busy:.2f}, lanes at work {sc['nodes_visited'] / max(1, 64 * sc['node_wave_steps']):.2f} / "
          f"{sc['prims_tested'] / max(1, 64 * sc['prim_wave_steps']):.2f}", flush=True)
