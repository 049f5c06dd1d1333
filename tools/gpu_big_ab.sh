# usage: gpu_big_ab.sh HALF SPP NAME...  -- the GPU tests, then synthetic_bvh:HALF with the tree's build and each variants/NAME
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/pytest_gpu.log 2>&1; rc=$?; echo exit=$rc; tail -4 gpurun_out/pytest_gpu.log | cut -c1-300; [ $rc -eq 0 ] || exit 1
half=$1; spp=$2; shift 2
for v in "" "$@"; do
  lib=${v:+$PWD/variants/$v/libmirt.so}
  MIRT_LIBMIRT=$lib timeout -k 10 600 python -u tools/big_env_sweep.py $half $spp MIRT_STEP_BUDGET 256 || exit 1
done
