#!/usr/bin/env python3
"""Times scene build/upload and a short render of the synthetic many-spheres scene (tables in global memory),
with the BVH built on the host and on the device."""
import os, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import raytracing_amd as R
half = int(sys.argv[1]) if len(sys.argv) > 1 else 500
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 8
for builder in (sys.argv[3:] or ["device", "host"]):
    os.environ["MIRT_BVH_BUILDER"] = builder
    t0 = time.perf_counter()
    g = R.ExampleScene(f"synthetic_bvh:{half}")
    t1 = time.perf_counter()
    print(f"synthetic_bvh:{half}, {builder}-built BVH: {g.primitive_count()} primitives, scene build + upload {t1 - t0:.2f} s", flush=True)
    g.render(1920, 1080, 1)
    _, _, st = g.render(1920, 1080, spp, flags=R.MIRT_RENDER_COUNT_WORK | R.MIRT_RENDER_TIME_KERNELS)
    seg = st["segments"]
    print(f"  1920x1080x{spp}: segments/sample {seg/(1920*1080*spp):.2f}, tree steps/seg {st['nodes_visited']/seg:.1f}, "
          f"primitive tests/seg {st['prims_tested']/seg:.2f}", flush=True)
    _, _, st = g.render(1920, 1080, spp, flags=R.MIRT_RENDER_TIME_KERNELS)
    print(f"  kernel {st['kernel_ms'][0]:.1f} ms = {1920*1080*spp/st['kernel_ms'][0]/1e3:.0f} Msamples/s", flush=True)
    g.close()
