# the headline frame (1024 spp) over the path pool: library default (32 Mi), 64 Mi, 128 Mi, default again
for pool in 0 67108864 134217728 0; do
  timeout -k 10 300 python -u bench.py --steps 3 --warmup 1 --pool $pool --no-cpu-baseline > gpurun_out/pool.json 2>gpurun_out/pool.err || { tail -3 gpurun_out/pool.err; exit 1; }
  python -c "
import json; d=json.load(open('gpurun_out/pool.json')); print('pool', d['config']['path_pool'], round(d['value'],1), round(d['ms_per_step'],2), round(d['roofline']['avg_launch_ms'],2), d['golden']['psnr_db'])"
done
