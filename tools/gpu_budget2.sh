set -e
R=$GRAFT_REPO_ROOT
timeout -k 10 300 python -u tools/big_budget_sweep.py 1581 8 64 128 256 512 1024 2>&1 | tee $R/gpurun_out/budget2_sweep_10m.log
timeout -k 10 300 python -u tools/big_budget_sweep.py 500 16 64 128 256 512 2>&1 | tee $R/gpurun_out/budget2_sweep_1m.log
timeout -k 10 200 python -u bench.py --steps 2 --warmup 1 --spp 256 --no-cpu-baseline | cut -c1-200
