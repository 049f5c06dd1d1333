#!/usr/bin/env python3
"""One plain render for counter collection: rocprofv3 --pmc <counters> -- python3 tools/pmc_probe.py [scene] [spp]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import raytracing_amd as R

name = sys.argv[1] if len(sys.argv) > 1 else "final_scene"
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 32
g = R.ExampleScene(name)
_, _, st = g.render(1920, 1080, spp, aspect=1920 / 1080)
print(name, spp, "spp", st["iterations"], "rounds", round(st["render_ms"], 2), "ms")
