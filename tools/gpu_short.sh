for spp in 1 2 4 8; do
  timeout -k 10 200 python -u bench.py --steps 4 --warmup 1 --spp $spp --no-cpu-baseline > gpurun_out/ab.json 2>gpurun_out/ab.err || { tail -3 gpurun_out/ab.err; exit 1; }
  python -c "
import json; d=json.load(open('gpurun_out/ab.json')); print($spp, 'spp', round(d['value'],1), 'ms/step', round(d['ms_per_step'],3), 'kernel', round(d['roofline']['avg_launch_ms'],3), 'rounds', d['config']['rounds_per_step'], 'pool', d['config']['path_pool'])"
done
