#!/usr/bin/env python3
"""Closest hits of a large synthetic scene against the oracle's exhaustive tournament (bvh.rs:92-145), bit for bit:
camera rays of the scene's own view (most of them cross hundreds of units of the field at a grazing angle, where the
f32 sphere test reports hits well off the spheres) and rays scattered over and above the field.
usage: big_parity_probe.py half [n_camera_columns]     (the oracle tests every sphere against every ray: minutes)"""
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
name = f"synthetic_bvh:{half}"
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


rays = np.concatenate([R.camera_rays(cam, cols, cols * 9 // 16),
                       scattered(600, [-half, 0.0, -half], [half, 2.0, half]),
                       scattered(300, [-1.2 * half, 0.01, -1.2 * half], [1.2 * half, 40.0, 1.2 * half])])
t0 = time.time()
ti, oi, t = g.trace(rays)
print(len(rays), "rays, device", round(time.time() - t0, 2), "s", flush=True)
t0 = time.time()
oti, ooi, ot = o.trace(rays)
print("oracle", round(time.time() - t0, 1), "s", flush=True)
hit = oti != 0xFFFFFFFF
far = np.linalg.norm(rays[:, 3:6], axis=1) * np.where(hit, ot, 0.0)
out = {"scene": name, "primitives": int(g.primitive_count()), "rays": int(len(rays)), "hits": int(hit.sum()),
       "hits_beyond_500_units": int((far > 500).sum()), "farthest_hit": float(far.max()),
       "ids_that_differ": int(((ti != oti) | (oi != ooi)).sum()),
       "t_bits_that_differ": int((t.view(np.uint32) != ot.view(np.uint32)).sum())}
print(json.dumps(out))
sys.exit(1 if out["ids_that_differ"] or out["t_bits_that_differ"] else 0)
