# final_scene (256 spp) over the walk's thresholds for scenes in LDS: MIRT_REFILL_MIN : MIRT_NODE_MIN
for rn in "32 16" "32 24" "40 16" "24 16" "24 8" "40 24" "32 16"; do
  set -- $rn
  MIRT_REFILL_MIN=$1 MIRT_NODE_MIN=$2 timeout -k 10 200 python -u bench.py --steps 2 --warmup 1 --spp 256 --no-cpu-baseline > gpurun_out/thr.json 2>gpurun_out/thr.err || { tail -3 gpurun_out/thr.err; exit 1; }
  python -c "
import json; d=json.load(open('gpurun_out/thr.json')); print('refill $1 node $2', round(d['value'],1), round(d['ms_per_step'],2), {k: round(v,3) for k,v in d['roofline']['wave_time_share'].items()}, {k: round(v,3) for k,v in d['roofline']['lanes_at_work'].items()})"
done
