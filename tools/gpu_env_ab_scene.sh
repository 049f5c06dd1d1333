# usage: gpu_env_ab_scene.sh VAR SCENE SPP  -- SCENE at 1080p without and with the environment switch VAR set, twice over
for round in 1 2; do
  for on in 0 1; do
    if [ $on = 1 ]; then export $1=1; else unset $1; fi
    timeout -k 10 200 python -u bench.py --scene $2 --steps 2 --warmup 1 --spp $3 --no-cpu-baseline > gpurun_out/ab.json 2>gpurun_out/ab.err || { tail -3 gpurun_out/ab.err; exit 1; }
    python -c "
import json; d=json.load(open('gpurun_out/ab.json')); print('$2 $1', $on, round(d['value'],1), round(d['ms_per_step'],2))"
  done
done
