for r in 8 16 24 32; do for n in 8 16 24; do
  MIRT_REFILL_MIN=$r MIRT_NODE_MIN=$n timeout -k 10 200 python -u bench.py --steps 2 --warmup 1 --spp 256 --no-cpu-baseline > gpurun_out/sweep.json 2>gpurun_out/sweep.err || exit 1
  python -c "
import json; d=json.load(open('gpurun_out/sweep.json')); print('refill $r node_min $n', round(d['value'],1))"
done; done
