#!/usr/bin/env python3
"""Throughput of the example scenes at their own resolution (1024 spp unless given), one GPU."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import raytracing_amd as R

spp = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
for name, w, h, aspect in (("spheres", 600, 337, 0.0), ("spheres_bvh", 600, 337, 0.0), ("cornel_box", 1000, 1000, 1.0),
                           ("final_scene", 1920, 1080, 1920 / 1080)):
    g = R.ExampleScene(name)
    g.render(w, h, 2, aspect=aspect)
    _, _, st = g.render(w, h, spp, aspect=aspect, flags=R.MIRT_RENDER_TIME_KERNELS)
    _, _, ws = g.render(w, h, 8, aspect=aspect, flags=R.MIRT_RENDER_COUNT_WORK)
    seg = ws["segments"]
    print(f"{name:12s} {w}x{h} {spp} spp depth {g.depth}: {st['kernel_ms'][0]:8.1f} ms = {w*h*spp/st['kernel_ms'][0]/1e3:7.0f} Msamples/s;"
          f" segments/sample {seg/(w*h*8):.2f}, tree steps/segment {ws['nodes_visited']/seg:.1f}, primitive tests/segment {ws['prims_tested']/seg:.2f}",
          flush=True)
    g.close()
