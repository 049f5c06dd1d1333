"""final_scene at the size of the render the reference's README shows (500x500), on the GPU, as PNGs under gpurun_out/.
usage: reference_image_probe.py SPP[:DEPTH] ..."""
import sys
from pathlib import Path

import numpy as np
from PIL import Image

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import raytracing_amd as R

out = ROOT / "gpurun_out"
out.mkdir(exist_ok=True)
g = R.ExampleScene("final_scene")
for arg in sys.argv[1:] or ["1024"]:
    spp, _, depth = arg.partition(":")
    spp, depth = int(spp), int(depth or 5)
    rgb, sums, stats = g.render(500, 500, spp, depth=depth, aspect=1.0)
    Image.fromarray(rgb).save(out / f"final_scene_500_{spp}spp_d{depth}.png")
    print(spp, "spp depth", depth, stats["iterations"], "iterations")
