# usage: gpu_big_variants.sh HALF SPP NAME...  -- the parity tests of the walk from global memory with the tree's
# build, then synthetic_bvh:HALF at SPP with each variants/NAME/libmirt.so and the tree's build, twice over
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_records.py tests/test_gpu_fuzz.py -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/pytest_walk.log 2>&1; rc=$?; tail -3 gpurun_out/pytest_walk.log | cut -c1-300; [ $rc -eq 0 ] || exit 1
half=$1; spp=$2; shift 2
for round in 1 2; do
  for v in "$@" ""; do
    lib=${v:+$PWD/variants/$v/libmirt.so}
    MIRT_LIBMIRT=$lib timeout -k 10 300 python -u tools/big_env_sweep.py $half $spp MIRT_STEP_BUDGET 256 || exit 1
  done
done
