# usage: gpu_ab_spp.sh SPP NAME...  -- bench at SPP samples per pixel: the tree's build and each variants/NAME, twice over
spp=$1; shift
for round in 1 2; do
  for v in "" "$@"; do
    lib=${v:+$PWD/variants/$v/libmirt.so}
    MIRT_LIBMIRT=$lib timeout -k 10 200 python -u bench.py --steps 4 --warmup 1 --spp $spp --no-cpu-baseline > gpurun_out/ab.json 2>gpurun_out/ab.err || { tail -3 gpurun_out/ab.err; exit 1; }
    python -c "
import json; d=json.load(open('gpurun_out/ab.json')); print('${v:-tree}', $spp, 'spp', round(d['value'],1), round(d['ms_per_step'],2), d['config']['rounds_per_step'])"
  done
done
