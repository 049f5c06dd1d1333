# usage: gpu_big_abc.sh HALF SPP ROUNDS NAME...  -- synthetic_bvh:HALF at SPP with each variants/NAME/libmirt.so and the
# tree's build in turn, ROUNDS times over, on one GPU box (kernel time by HIP events, best of 2 each)
half=$1; spp=$2; rounds=$3; shift 3
for round in $(seq $rounds); do
  for v in "$@" ""; do
    lib=${v:+$PWD/variants/$v/libmirt.so}
    MIRT_LIBMIRT=$lib timeout -k 10 300 python -u tools/big_spp_sweep.py $half $spp | sed "s|^ |${v:-tree} |" || exit 1
  done
done
