#!/usr/bin/env python3
"""Prints where the waves of the render kernel spend their cycles (instrumented variant, MIRT_RENDER_COUNT_WORK)."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import raytracing_amd as R

name = sys.argv[1] if len(sys.argv) > 1 else "final_scene"
w, h, spp = (1920, 1080, int(sys.argv[2]) if len(sys.argv) > 2 else 8)
g = R.ExampleScene(name)
g.render(w, h, 2, aspect=w / h)
_, _, st = g.render(w, h, spp, aspect=w / h, flags=R.MIRT_RENDER_COUNT_WORK | R.MIRT_RENDER_TIME_KERNELS)
c = st["phase_cycles"]
names = ["start+fetch", "tree steps", "primitive tests", "shade", "hand-in+bin", "total"]
seg = st["segments"]
print(f"{name}: segments {seg} nodes/seg {st['nodes_visited']/seg:.2f} prims/seg {st['prims_tested']/seg:.2f} "
      f"rounds {st['iterations']} kernel ms {st['kernel_ms'][0]:.2f}")
if st["node_wave_steps"]:
    print(f"  lanes at work: tree steps {st['nodes_visited']/(64*st['node_wave_steps']):.3f}  "
          f"primitive tests {st['prims_tested']/(64*max(1,st['prim_wave_steps'])):.3f}  "
          f"wave turns per 64 segments: {64*st['node_wave_steps']/seg:.1f} + {64*st['prim_wave_steps']/seg:.1f}")
for n, v in zip(names, c):
    print(f"  {n:16s} {v/1e9:9.3f} Gcycles  {100*v/max(1,c[5]):5.1f} %   {v/seg*64:8.1f} cycles per 64 segments")
