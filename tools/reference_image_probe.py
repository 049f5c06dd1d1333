"""final_scene at the size of the render the reference's README shows (500x500), on the GPU, as PNGs under gpurun_out/.
usage: reference_image_probe.py SPP[:DEPTH[:SCENE_SEED]] ..."""
import sys
from pathlib import Path

from PIL import Image

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import raytracing_amd as R

out = ROOT / "gpurun_out"
out.mkdir(exist_ok=True)
for arg in sys.argv[1:] or ["1024"]:
    parts = arg.split(":") + ["", ""]
    spp, depth, scene_seed = int(parts[0]), int(parts[1] or 5), int(parts[2] or 1)
    g = R.ExampleScene("final_scene", scene_seed=scene_seed)
    rgb, sums, stats = g.render(500, 500, spp, depth=depth, aspect=1.0)
    Image.fromarray(rgb).save(out / f"final_scene_500_{spp}spp_d{depth}_s{scene_seed}.png")
    print(spp, "spp depth", depth, "scene seed", scene_seed, stats["iterations"], "iterations")
    g.close()
