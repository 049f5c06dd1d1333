#!/usr/bin/env python3
"""Closest hits of a large synthetic scene against the oracle, bit for bit: camera rays of the scene's own view
(most of them cross hundreds of units of the field at a grazing angle, where the f32 sphere test reports hits well
off the spheres), rays scattered over and above the field, and rays that skim the field for up to 2.6 half-widths
before they meet a sphere.
usage: big_parity_probe.py half [n_camera_columns] [linear|bvh] [n_checker_rays]
  bvh (default): the oracle's bvh.rs tournament (bvh.rs:92-145; its build sorts at every level: slow beyond 1 M)
  linear:        the oracle's World::hit scan (world.rs:75-94) against the device's flat-World scene `synthetic:half`;
                 every sphere against every ray: minutes at 10 M
  n_checker_rays > 0: that many more rays against the oracle's culling container (oracle.hpp CulledBVH: the World's
                 own objects and tie rule behind a box test that culls), itself held to the linear scan on the rays above"""
import json
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import oracle_lib as O
import raytracing_amd as R

half = int(sys.argv[1]) if len(sys.argv) > 1 else 500
cols = int(sys.argv[2]) if len(sys.argv) > 2 else 48
mode = sys.argv[3] if len(sys.argv) > 3 else "bvh"
n_checker = int(sys.argv[4]) if len(sys.argv) > 4 else 0
name = f"synthetic_bvh:{half}" if mode == "bvh" else f"synthetic:{half}"
t0 = time.time()
g = R.ExampleScene(name)
print(name, g.primitive_count(), "primitives, device scene", round(time.time() - t0, 1), "s", flush=True)
t0 = time.time()
o = O.Scene(name)
print("oracle scene", round(time.time() - t0, 1), "s", flush=True)
cam, _ = g.camera()
rng = np.random.default_rng(5)
F = np.float32


def scattered(n, lo, hi):
    rays = np.zeros((n, 7), F)
    rays[:, 0:3] = rng.uniform(lo, hi, (n, 3))
    d = rng.normal(size=(n, 3))
    d[:, 1] *= 0.05                                         # mostly along the field
    rays[:, 3:6] = d * rng.uniform(0.2, 3.0, (n, 1))
    rays[:, 6] = rng.uniform(0, 1, n)
    return rays


def skimming(n):
    rays = np.zeros((n, 7), F)
    angle = rng.uniform(0, 2 * np.pi, n)
    start = rng.uniform(0.95, 1.3, n) * half
    rays[:, 0] = -np.cos(angle) * start + rng.uniform(-0.3, 0.3, n) * half
    rays[:, 2] = -np.sin(angle) * start + rng.uniform(-0.3, 0.3, n) * half
    rays[:, 1] = rng.uniform(0.45, 3.0, n)
    reach = rng.uniform(0.8, 2.6, n) * half
    scale = rng.uniform(0.2, 3.0, n)
    rays[:, 3] = np.cos(angle) * scale
    rays[:, 5] = np.sin(angle) * scale
    rays[:, 4] = -(rays[:, 1] - 0.4) / reach * scale
    rays[:, 6] = rng.uniform(0, 1, n)
    return rays


def some_rays(n_cols, n_in, n_over, n_skim):
    return np.concatenate([R.camera_rays(cam, n_cols, n_cols * 9 // 16),
                           scattered(n_in, [-half, 0.0, -half], [half, 2.0, half]),
                           scattered(n_over, [-1.2 * half, 0.01, -1.2 * half], [1.2 * half, 40.0, 1.2 * half]),
                           skimming(n_skim)])


rays = some_rays(cols, 400, 200, 500)
t0 = time.time()
ti, oi, t = g.trace(rays)
print(len(rays), "rays, device", round(time.time() - t0, 2), "s", flush=True)
t0 = time.time()
oti, ooi, ot = o.trace(rays)
print("oracle", round(time.time() - t0, 1), "s", flush=True)
hit = oti != 0xFFFFFFFF
far = np.linalg.norm(rays[:, 3:6], axis=1) * np.where(hit, ot, 0.0)
out = {"scene": name, "oracle": "World::hit linear scan (world.rs:75-94)" if mode == "linear" else "bvh.rs tournament",
       "primitives": int(g.primitive_count()), "rays": int(len(rays)), "hits": int(hit.sum()),
       "hits_beyond_500_units": int((far > 500).sum()), "hits_beyond_1000_units": int((far > 1000).sum()),
       "hits_beyond_1500_units": int((far > 1500).sum()), "farthest_hit": float(far.max()),
       "ids_that_differ": int(((ti != oti) | (oi != ooi)).sum()),
       "t_bits_that_differ": int((t.view(np.uint32) != ot.view(np.uint32)).sum())}
bad = out["ids_that_differ"] or out["t_bits_that_differ"]
if n_checker > 0 and mode == "linear":
    t0 = time.time()
    c = O.Scene(f"synthetic_culled:{half}")
    print("culling container", round(time.time() - t0, 1), "s", flush=True)
    cti, coi, ct = c.trace(rays)
    out["culling_container_vs_linear_scan"] = {"ids_that_differ": int(((cti != oti) | (coi != ooi)).sum()),
                                               "t_bits_that_differ": int((ct.view(np.uint32) != ot.view(np.uint32)).sum())}
    more = some_rays(max(cols, int((n_checker // 2 * 16 / 9) ** 0.5)), n_checker // 8, n_checker // 8, n_checker // 4)
    ti, oi, t = g.trace(more)
    cti, coi, ct = c.trace(more)
    hit = cti != 0xFFFFFFFF
    far = np.linalg.norm(more[:, 3:6], axis=1) * np.where(hit, ct, 0.0)
    out["device_vs_culling_container"] = {"rays": int(len(more)), "hits": int(hit.sum()),
                                          "hits_beyond_1000_units": int((far > 1000).sum()),
                                          "hits_beyond_1500_units": int((far > 1500).sum()), "farthest_hit": float(far.max()),
                                          "ids_that_differ": int(((ti != cti) | (oi != coi)).sum()),
                                          "t_bits_that_differ": int((t.view(np.uint32) != ct.view(np.uint32)).sum())}
    bad = bad or any(out[k][j] for k in ("culling_container_vs_linear_scan", "device_vs_culling_container")
                     for j in ("ids_that_differ", "t_bits_that_differ"))
print(json.dumps(out))
(ROOT / "gpurun_out").mkdir(exist_ok=True)
(ROOT / "gpurun_out" / f"big_parity_probe_{half}_{mode}.json").write_text(json.dumps(out, indent=1))
sys.exit(1 if bad else 0)
