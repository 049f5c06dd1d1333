# throughput against samples per pixel and pool size: usage gpu_pool_spp.sh "SPP..." "POOL..."
for spp in $1; do
  for pool in $2; do
    timeout -k 10 200 python -u bench.py --steps 3 --warmup 1 --spp $spp --pool $pool --no-cpu-baseline > gpurun_out/ps.json 2>gpurun_out/ps.err || { tail -3 gpurun_out/ps.err; exit 1; }
    python -c "
import json; d=json.load(open('gpurun_out/ps.json')); print('spp', $spp, 'pool', $pool, round(d['value'],1), 'Msamples/s', round(d['ms_per_step'],2), 'ms/step', 'kernel', round(d['roofline']['avg_launch_ms'],2), 'rounds', d['config']['rounds_per_step'])"
  done
done
