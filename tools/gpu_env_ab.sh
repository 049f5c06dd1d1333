# usage: gpu_env_ab.sh VAR VALUE...   -- bench (256 spp) the library with VAR set to each VALUE in turn, twice over, on one GPU box
var=$1; shift
for round in 1 2; do
  for v in "$@"; do
    env $var=$v timeout -k 10 200 python -u bench.py --steps 2 --warmup 1 --spp 256 --no-cpu-baseline > gpurun_out/ab.json 2>gpurun_out/ab.err || { tail -3 gpurun_out/ab.err; exit 1; }
    python -c "
import json; d=json.load(open('gpurun_out/ab.json')); r=d['roofline']; print('$var=$v', round(d['value'],1), round(d['ms_per_step'],2), 'n', round(r['nodes_per_segment'],3), 'p', round(r['prims_per_segment'],3), {k: round(v,3) for k,v in r['wave_time_share'].items()})"
  done
done
