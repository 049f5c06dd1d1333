#!/usr/bin/env python3
"""Generates the fixtures under tests/golden/ with the CPU oracle (oracle/liboracle.so).

The reference ships no golden vectors for this path and cannot be built here, so the
fixtures are outputs of the oracle (the C++ restatement of the reference), made once in
the build container:

  spheres_bvh_primary.npz   closest-hit (type_index, object_index, t bits) of the 600x337
                            pixel-centre rays of examples/spheres_bvh through the oracle's
                            bvh.rs traversal, scene seed 1
  <scene>_converged.npz     low-resolution, high-spp renders (RGB24 + mask of NaN-poisoned pixels + spp +
                            seed) used as the converged CPU image for PSNR checks

  <scene>_blocks.npz        the frame the benchmark times, at the benchmark's own resolution and camera, rendered
                            by the oracle at `spp` samples per pixel and averaged over 8x8 pixel blocks: per block
                            the mean linear radiance of its finite pixels (f32, so spp * 64 samples per value) and
                            the number of pixels a NaN sample poisoned.  A GPU frame of the same resolution is
                            reduced the same way and compared block by block.

Usage: python tools/make_goldens.py [primary] [spheres] [cornel_box] [final_scene] [final_scene_blocks] [cornel_box_blocks]
"""
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "tests"))
import oracle_lib as O  # noqa: E402

GOLD = ROOT / "tests" / "golden"
GOLD.mkdir(exist_ok=True)
SEED = 1234
CONVERGED = {  # name: (width, height, spp)
    "spheres": (96, 54, 4096),
    "cornel_box": (64, 64, 16384),
    "final_scene": (64, 36, 16384),
}

BLOCKS = {  # name: (scene, width, height, spp per pixel, block edge)
    "final_scene_blocks": ("final_scene", 1920, 1080, 256, 8),
    "cornel_box_blocks": ("cornel_box", 1000, 1000, 256, 8),
}


def block_means(sums, spp, block):
    """Mean linear radiance over the finite pixels of each block x block tile, and the poisoned-pixel count."""
    h, w, _ = sums.shape
    finite = np.isfinite(sums).all(axis=2)
    lin = np.where(finite[..., None], sums.astype(np.float64) / spp, 0.0)
    hb, wb = h // block, w // block
    lin = lin[:hb * block, :wb * block].reshape(hb, block, wb, block, 3).sum(axis=(1, 3))
    cnt = finite[:hb * block, :wb * block].reshape(hb, block, wb, block).sum(axis=(1, 3))
    mean = lin / np.maximum(cnt, 1)[..., None]
    return mean.astype(np.float32), (block * block - cnt).astype(np.uint8)


what = sys.argv[1:] or ["primary"] + list(CONVERGED)
for name, (scene, w, h, spp, block) in BLOCKS.items():
    if name not in what:
        continue
    t0 = time.time()
    s = O.Scene(scene, scene_seed=1)
    threads = max(1, (len(__import__("os").sched_getaffinity(0))) - 1)
    _, sums, seg = s.render(w, h, spp, aspect=w / h, seed=SEED, threads=threads)
    mean, poisoned = block_means(sums, spp, block)
    np.savez_compressed(GOLD / f"{name}.npz", mean=mean, poisoned=poisoned, width=w, height=h, spp=spp, block=block,
                        seed=SEED, scene_seed=1, depth=s.depth, aspect=np.float32(w / h),
                        segments_per_sample=np.float64(seg / (w * h * spp)))
    print(name, w, h, spp, round(time.time() - t0, 1), "s", flush=True)
if "primary" in what:
    s = O.Scene("spheres_bvh", scene_seed=1)
    cam, _ = s.camera()
    rays = O.pixel_centre_rays(cam, 600, 337)
    ti, oi, t = s.trace(rays)
    np.savez_compressed(GOLD / "spheres_bvh_primary.npz", type_index=ti.astype(np.uint8), object_index=oi.astype(np.uint16),
                        hit=(ti != 0xFFFFFFFF), t_bits=t.view(np.uint32), width=600, height=337, scene_seed=1)
    print("primary", np.mean(ti != 0xFFFFFFFF))
for name, (w, h, spp) in CONVERGED.items():
    if name not in what:
        continue
    t0 = time.time()
    s = O.Scene(name, scene_seed=1)
    rgb, sums, seg = s.render(w, h, spp, aspect=w / h, seed=SEED)
    # a NaN sample blackens its pixel (renderer.rs:241-252); which pixels get one depends on the random
    # streams, so they are recorded and left out of PSNR comparisons
    nan_mask = ~np.isfinite(sums).all(axis=2)
    np.savez_compressed(GOLD / f"{name}_converged.npz", rgb=rgb, nan_mask=nan_mask, width=w, height=h, spp=spp, seed=SEED, scene_seed=1,
                        depth=s.depth, aspect=np.float32(w / h))
    print(name, w, h, spp, round(time.time() - t0, 1), "s", flush=True)
