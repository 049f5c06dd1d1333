# usage: gpu_big_pool_sweep.sh SPP POOL...  -- the 10 M-sphere scene at 1080p, SPP samples, for each pool size (0: the library's own)
spp=$1; shift
mkdir -p gpurun_out
for pool in "$@"; do
  timeout -k 10 500 python -u bench.py --scene synthetic_bvh:1581 --spp $spp --pool $pool --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/c5.json 2> gpurun_out/c5.err || { tail -3 gpurun_out/c5.err; exit 1; }
  python -c "
import json; d=json.load(open('gpurun_out/c5.json')); r=d['roofline']; print('spp $spp pool $pool:', round(d['value'],1), 'Msamples/s', round(d['ms_per_step'],2), 'ms', d['config']['rounds_per_step'], 'rounds')"
done
