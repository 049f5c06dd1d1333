# usage: gpu_ab_scene.sh "BENCH ARGS" NAME...   -- bench.py with BENCH ARGS on each variants/NAME/libmirt.so in turn, twice over
args=$1; shift
for round in 1 2; do
  for v in "$@"; do
    MIRT_LIBMIRT=$PWD/variants/$v/libmirt.so timeout -k 10 300 python -u bench.py --steps 2 --warmup 1 --no-cpu-baseline $args > gpurun_out/ab.json 2>gpurun_out/ab.err || { tail -3 gpurun_out/ab.err; exit 1; }
    python -c "
import json; d=json.load(open('gpurun_out/ab.json')); r=d['roofline']; print('$v', d['config']['workload'], round(d['value'],1), round(d['ms_per_step'],2), 'n', round(r['nodes_per_segment'],3), 'p', round(r['prims_per_segment'],3))"
  done
done
