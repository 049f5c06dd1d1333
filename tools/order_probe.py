#!/usr/bin/env python3
"""What the order of second-bounce rays is worth to the walk: rays leaving the first hits of final_scene's camera
rays in random cosine-weighted directions, traced (mirt_trace_device, the render kernel's own walk) in several orders."""
import ctypes as C
import sys
import time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import torch
import raytracing_amd as R

F = np.float32
w, h = 1920, 1080
g = R.ExampleScene("final_scene")
L = R.mirt()
cam, _ = g.camera(w / h)
rays = R.camera_rays(cam, w, h)
# camera rays in the renderer's order: 8x8 pixel tiles, tile after tile
ys, xs = np.divmod(np.arange(w * h), w)
tile_key = (ys // 8) * (w // 8 + 1) * 64 + (xs // 8) * 64 + (ys % 8) * 8 + xs % 8
rays = rays[np.argsort(tile_key, kind="stable")]
rec = R.trace_records(g, rays)
hit = np.isfinite(rec["t"]) & (rec["t"] > 0)
rng = np.random.default_rng(5)
copies = 16                                                   # sixteen samples per pixel, sample after sample
n1 = int(hit.sum())
n = n1 * copies
second = np.zeros((n, 7), F)
for c in range(copies):
    u = rng.normal(size=(n1, 3)); u /= np.linalg.norm(u, axis=1, keepdims=True)
    second[c * n1:(c + 1) * n1, 0:3] = rec["point"][hit]
    second[c * n1:(c + 1) * n1, 3:6] = (rec["normal"][hit] + u).astype(F)   # cosine-weighted around the normal
second[:, 6] = 0.5
ctx, scene = g.L.rt_example_context(g.h), g.L.rt_example_scene(g.h)


def run(name, order):
    r = torch.from_numpy(np.ascontiguousarray(second[order])).cuda()
    ti = torch.zeros(n, dtype=torch.int32, device="cuda"); oi = torch.zeros_like(ti)
    t = torch.zeros(n, dtype=torch.float32, device="cuda")
    best = None
    for _ in range(5):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        rc = L.mirt_trace_device(ctx, scene, C.c_void_p(r.data_ptr()), n, 7, C.c_void_p(ti.data_ptr()),
                                 C.c_void_p(oi.data_ptr()), C.c_void_p(t.data_ptr()), None)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        assert rc == 0
        best = dt if best is None else min(best, dt)
    print(f"{name:46s} {best * 1e3:7.3f} ms  {n / best / 1e9:6.2f} G rays/s", flush=True)


octant = (second[:, 3] > 0).astype(int) * 4 + (second[:, 4] > 0).astype(int) * 2 + (second[:, 5] > 0).astype(int)
idx = np.arange(n)
block = idx // 2048
run("tile order (as the first hits come)", idx)
run("shuffled within blocks of 2048", np.lexsort((rng.random(n), block)))
run("shuffled within blocks of 65536", np.lexsort((rng.random(n), idx // 65536)))
run("shuffled altogether", rng.permutation(n))
run("tile order, by octant within blocks of 2048", np.lexsort((idx, octant, block)))
run("tile order, by octant within blocks of 8192", np.lexsort((idx, octant, idx // 8192)))
sh = np.lexsort((rng.random(n), block))
run("shuffled in 2048, then by octant in 2048", sh[np.lexsort((np.arange(n), octant[sh], np.arange(n) // 2048))])
cell = (np.floor(second[:, 0] / 70).astype(int) * 64 + np.floor(second[:, 2] / 70).astype(int))
run("shuffled in 8192, then by cell and octant", (lambda s8: s8[np.lexsort((np.arange(n), octant[s8], cell[s8], np.arange(n) // 8192))])(np.lexsort((rng.random(n), idx // 8192))))
