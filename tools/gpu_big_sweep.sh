# usage: gpu_big_sweep.sh HALF SPP VAR v1 v2 ...  -- synthetic_bvh:HALF at 1080p, SPP samples per pixel, for each value of VAR
half=$1; spp=$2; var=$3; shift 3
for v in "$@"; do
  env $var=$v MIRT_BVH_BUILDER=device timeout -k 10 300 python -u tools/big_scene_probe.py $half $spp device 2>&1 | tail -1 | sed "s/^/$var=$v /"
done
