# One GPU's share of config 5 (3840x2160, SPP samples) under a few settings of the walk's switches.
# usage: gpu_config5_env.sh SPP "VAR=value ..." ["VAR=value ..." ...]     ("" = the defaults)
spp=$1; shift
for setting in "$@"; do
  env $setting timeout -k 10 500 python -u bench.py --scene synthetic_bvh:1581 --width 3840 --height 2160 --spp $spp --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/c5.json 2> gpurun_out/c5.err || { tail -3 gpurun_out/c5.err; exit 1; }
  python -c "
import json; d=json.load(open('gpurun_out/c5.json')); r=d['roofline']; print('[$setting]', round(d['value'],1), 'Msamples/s', d['config']['rounds_per_step'], 'rounds', d['config']['path_pool'], 'pool', {k: round(v,2) for k,v in r['lanes_at_work'].items()}, {k: round(v,2) for k,v in r['wave_time_share'].items()})"
done
