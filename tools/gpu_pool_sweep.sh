for v in "$@"; do
  timeout -k 10 200 python -u bench.py --steps 2 --warmup 1 --spp 256 --no-cpu-baseline --pool $v > gpurun_out/sweep.json 2>gpurun_out/sweep.err || exit 1
  python -c "
import json; d=json.load(open('gpurun_out/sweep.json')); print('pool=$v', round(d['value'],1), round(d['ms_per_step'],2), {k: round(v,3) for k,v in d['roofline']['wave_time_share'].items()})"
done
