# usage: gpu_big_only.sh HALF SPP NAME...  -- synthetic_bvh:HALF at SPP with each variants/NAME/libmirt.so, twice over (no tests)
half=$1; spp=$2; shift 2
for round in 1 2; do
  for v in "$@"; do
    MIRT_LIBMIRT=$PWD/variants/$v/libmirt.so timeout -k 10 300 python -u tools/big_env_sweep.py $half $spp MIRT_STEP_BUDGET 256 || exit 1
  done
done
