# usage: gpu_long_ab.sh SPP NAME...  -- the 10 M-sphere scene, 1080p, SPP samples, 32 Mi pool: the tree's build and each variants/NAME
spp=$1; shift
for v in "" "$@"; do
  MIRT_LIBMIRT=${v:+$PWD/variants/$v/libmirt.so} timeout -k 10 500 python -u bench.py --scene synthetic_bvh:1581 --spp $spp --pool 33554432 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/c5.json 2> gpurun_out/c5.err || { tail -3 gpurun_out/c5.err; exit 1; }
  python -c "
import json; d=json.load(open('gpurun_out/c5.json')); r=d['roofline']; print('[${v:-tree}]', round(d['value'],1), 'Msamples/s', round(d['ms_per_step'],1), 'ms', d['config']['rounds_per_step'], 'rounds', 'n', round(r['nodes_per_segment'],1), 'p', round(r['prims_per_segment'],1))"
done
