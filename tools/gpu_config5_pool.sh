# usage: gpu_config5_pool.sh  -- the 10 M-sphere scene: frame size x pool size
for cfg in "3840 2160 128 8388608" "3840 2160 128 33554432" "1920 1080 512 33554432" "1920 1080 512 8388608" "1920 1080 64 8388608"; do
  set -- $cfg
  timeout -k 10 500 python -u bench.py --scene synthetic_bvh:1581 --width $1 --height $2 --spp $3 --pool $4 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/c5.json 2> gpurun_out/c5.err || { tail -3 gpurun_out/c5.err; exit 1; }
  python -c "
import json; d=json.load(open('gpurun_out/c5.json')); r=d['roofline']; print('[$cfg]', round(d['value'],1), 'Msamples/s', d['config']['rounds_per_step'], 'rounds', 'n', round(r['nodes_per_segment'],1), 'p', round(r['prims_per_segment'],1), 's', round(r['segments_per_sample'],2), {k: round(v,2) for k,v in r['lanes_at_work'].items()})"
done
