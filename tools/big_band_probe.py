#!/usr/bin/env python3
"""Where in the frame the long walks of a large scene are: the frame rendered band of rows by band of rows (the row
window of mirt_render), kernel time and rounds of each band at a small and a larger sample count.
usage: big_band_probe.py half [rows_per_band=120]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import raytracing_amd as R

half = int(sys.argv[1])
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 120
name = f"synthetic_bvh:{half}"
ex = R.ExampleScene(name)
cam, bg = ex.camera(1920 / 1080)
depth = ex.depth
s = R.SharedScene(name, [0])
W, H = 1920, 1080
s.render(cam, bg, W, H, 1, depth, want_sums=False)
for first in range(0, H, rows):
    n = min(rows, H - first)
    out = []
    for spp in (4, 32):
        best = None
        for _ in range(2):
            _, _, st = s.render(cam, bg, W, H, spp, depth, first_row=first, n_rows=n, flags=R.MIRT_RENDER_TIME_KERNELS, want_sums=False)
            ms = st["kernel_ms"][0]
            if best is None or ms < best:
                best, rounds = ms, st["iterations"]
        out.append(f"{spp} spp {best:7.2f} ms {rounds:3d} rounds")
    print(f"rows {first:4d}..{first + n - 1:4d}: " + "; ".join(out), flush=True)
