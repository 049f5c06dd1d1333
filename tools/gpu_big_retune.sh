# Large scenes after a change of the walk: budgets and thresholds at 10 M spheres, the 1 M and 6.5 k rows, and one
# GPU's share of config 5 at its own size.   usage: gpu_big_retune.sh TAG
set -e
O=gpurun_out/$1; mkdir -p $O
timeout -k 10 400 python -u tools/big_env_sweep.py 1581 64 MIRT_STEP_BUDGET 128 256 512 1024 | tee $O/budget_10m.txt
timeout -k 10 400 env MIRT_STEP_BUDGET=256 python -u tools/big_env_sweep.py 1581 64 MIRT_NODE_MIN 8 16 24 32 | tee $O/node_min_10m.txt
timeout -k 10 400 env MIRT_STEP_BUDGET=256 python -u tools/big_env_sweep.py 1581 64 MIRT_REFILL_MIN 16 32 48 | tee $O/refill_min_10m.txt
timeout -k 10 300 python -u tools/big_env_sweep.py 500 32 MIRT_STEP_BUDGET 128 256 512 | tee $O/budget_1m.txt
timeout -k 10 300 python -u tools/big_env_sweep.py 40 256 MIRT_STEP_BUDGET 0 256 | tee $O/budget_6k.txt
timeout -k 10 500 python -u bench.py --scene synthetic_bvh:1581 --width 3840 --height 2160 --spp 512 --steps 2 --warmup 1 --no-cpu-baseline > $O/config5_share.json 2> $O/config5_share.err
cat $O/config5_share.json | cut -c1-400
