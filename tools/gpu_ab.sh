# usage: gpu_ab.sh NAME...   -- bench (256 spp) each variants/NAME/libmirt.so in turn, twice over, on one GPU box
for round in 1 2; do
  for v in "$@"; do
    MIRT_LIBMIRT=$PWD/variants/$v/libmirt.so timeout -k 10 200 python -u bench.py --steps 2 --warmup 1 --spp 256 --no-cpu-baseline > gpurun_out/ab.json 2>gpurun_out/ab.err || { tail -3 gpurun_out/ab.err; exit 1; }
    python -c "
import json; d=json.load(open('gpurun_out/ab.json')); print('$v', round(d['value'],1), round(d['ms_per_step'],2), {k: round(v,3) for k,v in d['roofline']['wave_time_share'].items()})"
  done
done
