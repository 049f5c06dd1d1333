# usage: gpu_ab_test.sh [SPP] NAME...  -- the GPU tests, then the tree's build against variants/NAME at 256 spp, twice over
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/pytest_gpu.log 2>&1; rc=$?; echo exit=$rc; tail -5 gpurun_out/pytest_gpu.log | cut -c1-300; [ $rc -eq 0 ] || exit 1
for round in 1 2; do
  for v in "$@" ""; do
    lib=${v:+$PWD/variants/$v/libmirt.so}
    MIRT_LIBMIRT=$lib timeout -k 10 200 python -u bench.py --steps 2 --warmup 1 --spp 256 --no-cpu-baseline > gpurun_out/ab.json 2>gpurun_out/ab.err || { tail -3 gpurun_out/ab.err; exit 1; }
    python -c "
import json; d=json.load(open('gpurun_out/ab.json')); print('${v:-tree}', round(d['value'],1), round(d['ms_per_step'],2), d['config']['rounds_per_step'], {k: round(v,3) for k,v in d['roofline']['wave_time_share'].items()})"
  done
done
