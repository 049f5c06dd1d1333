#!/usr/bin/env python3
"""One plain render of a large synthetic scene for counter collection (rocprofv3 --pmc ... -- python3 this half spp)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import raytracing_amd as R

half = int(sys.argv[1]) if len(sys.argv) > 1 else 1581
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 8
flags = R.MIRT_RENDER_COUNT_WORK if len(sys.argv) > 3 and sys.argv[3] == "count" else 0
g = R.ExampleScene(f"synthetic_bvh:{half}")
_, _, st = g.render(1920, 1080, spp, aspect=1920 / 1080, flags=flags)
print(f"synthetic_bvh:{half}", g.primitive_count(), "primitives", spp, "spp", st["iterations"], "rounds",
      round(st["render_ms"], 2), "ms", "segments", st["segments"], "nodes", st["nodes_visited"], "prims", st["prims_tested"])
