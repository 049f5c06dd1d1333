# What the reference's README frame can tell apart: builds a copy of the device library with one of the reference's
# rules changed and renders the frame with it (on the GPU box: run the second half there).
#   here:        bash tools/reference_image_variants.sh build ln|metal_absorbs|medium_normal_flipped
#   on the box:  bash tools/reference_image_variants.sh render NAME   -> gpurun_out/final_scene_500_NAME.png
# Region means of such a frame against tests/golden/reference_final_scene_500x500.npz: tests/test_reference_image.py's
# discs (the figures quoted for these variants were taken at 2048 spp, depth 50).
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
mode=$1; name=$2
if [ "$mode" = build ]; then
  tmp=$(mktemp -d)
  mkdir -p "$tmp/raytracing_amd" "$root/variants/$name"
  cp -r "$root/raytracing_amd/csrc" "$tmp/raytracing_amd/"; cp -r "$root/include" "$tmp/"
  k="$tmp/raytracing_amd/csrc/kernels.hip"
  case "$name" in
    ln) sed -i 's/med.neg_inv_density \* log2f(rng_at(rng, m))/med.neg_inv_density * logf(rng_at(rng, m))/' "$k" ;;
    metal_absorbs) python3 - "$k" <<'PY'
import sys
p = sys.argv[1]; s = open(p).read()
old = "nr.o = h.p; nr.d = reflected + fuzz; nr.time = r.time;\n            alive = true;"
assert old in s
open(p, "w").write(s.replace(old, "nr.o = h.p; nr.d = reflected + fuzz; nr.time = r.time;\n            alive = dot(nr.d, h.n) > 0.0f;", 1))
PY
      ;;
    medium_normal_flipped) python3 - "$k" <<'PY'
import sys
p = sys.argv[1]; s = open(p).read()
old = "h.p = ray_at(r, t_found);\n    h.n = v3(1.0f, 0.0f, 0.0f);"
assert old in s
open(p, "w").write(s.replace(old, "h.p = ray_at(r, t_found);\n    h.n = v3(-1.0f, 0.0f, 0.0f);", 1))
PY
      ;;
    *) echo "unknown variant $name"; exit 1 ;;
  esac
  cmp -s "$k" "$root/raytracing_amd/csrc/kernels.hip" && { echo "the pattern of $name no longer matches kernels.hip"; exit 1; }
  make -s -C "$tmp/raytracing_amd/csrc" OUT="$root/variants/$name/libmirt.so"
  rm -rf "$tmp"
  echo "built variants/$name/libmirt.so"
else
  MIRT_LIBMIRT=$root/variants/$name/libmirt.so timeout -k 10 200 python3 -u "$root/tools/reference_image_probe.py" 2048:50:1
  mv "$root/gpurun_out/final_scene_500_2048spp_d50_s1.png" "$root/gpurun_out/final_scene_500_$name.png"
fi
