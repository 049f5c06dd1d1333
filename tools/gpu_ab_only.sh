# usage: gpu_ab_only.sh NAME...  -- bench (256 spp) the tree's build and each variants/NAME, no tests
for v in "" "$@"; do
  lib=${v:+$PWD/variants/$v/libmirt.so}
  MIRT_LIBMIRT=$lib timeout -k 10 200 python -u bench.py --steps 2 --warmup 1 --spp 256 --no-cpu-baseline > gpurun_out/ab.json 2>gpurun_out/ab.err || { tail -3 gpurun_out/ab.err; exit 1; }
  python -c "
import json; d=json.load(open('gpurun_out/ab.json')); r=d['roofline']; print('${v:-tree}', round(d['value'],1), round(d['ms_per_step'],2), 'n/seg', round(r['nodes_per_segment'],2), 'p/seg', round(r['prims_per_segment'],2), 'seg/sample', round(r['segments_per_sample'],3), {k: round(v,3) for k,v in r['wave_time_share'].items()})"
done
