#!/usr/bin/env python3
"""Ad-hoc GPU check: primary-hit parity and same-seed image comparison against the oracle."""
import sys, time, json
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import raytracing_amd as R
import oracle_lib as O
from PIL import Image

out = ROOT / "gpurun_out"; out.mkdir(exist_ok=True)
res = {}
for name in ["spheres_bvh", "spheres", "cornel_box", "final_scene"]:
    t0 = time.time()
    g = R.ExampleScene(name); o = O.Scene(name)
    print(name, "open", round(time.time() - t0, 2), "s prims", g.primitive_count(), flush=True)
    w, h = (600, 337) if name.startswith("spheres") else (200, 200)
    cam, bg = g.camera()
    ocam, obg = o.camera()
    gc = np.concatenate([np.array(list(getattr(cam, k)), np.float32) for k in ["origin","lower_left_corner","horizontal","vertical","u","v","w"]])
    print(" camera equal:", np.array_equal(gc, ocam[:21]), flush=True)
    rays = R.camera_rays(cam, w, h)
    ti, oi, t = g.trace(rays)
    oti, ooi, ot = o.trace(rays)
    same_id = (ti == oti) & (oi == ooi)
    same_t = (t.view(np.uint32) == ot.view(np.uint32))
    print(f" trace: ids equal {same_id.mean():.6f} t bits equal {same_t.mean():.6f} hits {np.mean(oti != 0xFFFFFFFF):.3f}", flush=True)
    res[name] = {"ids": float(same_id.mean()), "tbits": float(same_t.mean())}
    if not same_id.all():
        bad = np.nonzero(~same_id)[0][:5]
        for b in bad: print("  bad", b, ti[b], oi[b], t[b], "oracle", oti[b], ooi[b], ot[b])
    rw, rh, spp = (150, 84, 16) if name.startswith("spheres") else (96, 96, 32)
    t0 = time.time(); rgb, sums, st = g.render(rw, rh, spp, flags=R.MIRT_RENDER_COUNT_WORK); tg = time.time() - t0
    t0 = time.time(); orgb, osums, oseg = o.render(rw, rh, spp); to = time.time() - t0
    Image.fromarray(rgb).save(out / f"{name}_gpu.png"); Image.fromarray(orgb).save(out / f"{name}_oracle.png")
    diff = np.abs(rgb.astype(int) - orgb.astype(int))
    print(f" render {rw}x{rh}x{spp}: gpu {tg:.2f}s oracle {to:.2f}s psnr {O.psnr(rgb, orgb):.2f} dB  max diff {diff.max()} frac>2 {np.mean(diff>2):.4f} seg gpu {st['segments']} oracle {oseg} iters {st['iterations']}", flush=True)
    res[name].update({"psnr": O.psnr(rgb, orgb), "seg_gpu": st["segments"], "seg_oracle": oseg})
(out / "gpu_check.json").write_text(json.dumps(res, indent=1))
