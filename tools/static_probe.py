#!/usr/bin/env python3
"""Rate of mirt_render_accumulate_static (fixed workspace) beside mirt_render_accumulate on a 1080p final_scene frame."""
import ctypes as C
import sys
import time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import raytracing_amd as R

spp = int(sys.argv[1]) if len(sys.argv) > 1 else 8
L = R.mirt()
g = R.ExampleScene("final_scene")
w, h = 1920, 1080
cam, bg = g.camera(w / h)
sums = torch.zeros((h, w, 3), dtype=torch.float32, device="cuda:0")
ctx, scene = g.L.rt_example_context(g.h), g.L.rt_example_scene(g.h)
for name, entry in (("allocating", L.mirt_render_accumulate), ("static", L.mirt_render_accumulate_static)):
    p = R.RenderParams(w, h, spp, 0, g.depth, (C.c_float * 3)(*bg), 11, 0, 0)
    st = R.Stats()
    for _ in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        rc = entry(ctx, scene, C.byref(cam), C.byref(p), C.c_void_p(sums.data_ptr()), None, C.byref(st))
        dt = time.perf_counter() - t0
        assert rc == 0, L.mirt_last_error(ctx).decode()
    print(f"{name:10s} {spp} spp: {dt * 1e3:8.2f} ms = {w * h * spp / dt / 1e6:7.1f} Msamples/s, pool {st.as_dict()['path_pool']}, "
          f"{st.as_dict()['iterations']} rounds", flush=True)
