set -e
R=$GRAFT_REPO_ROOT
timeout -k 10 300 python -u -m pytest tests/test_gpu_parity.py -x -q --timeout 200 --timeout-method thread -m gpu -k "put_aside or large_scene" > $R/gpurun_out/budget_tests.log 2>&1 || { tail -30 $R/gpurun_out/budget_tests.log; exit 1; }
tail -2 $R/gpurun_out/budget_tests.log
timeout -k 10 300 python -u tools/big_budget_sweep.py 1581 8 0 64 128 256 512 1024 2048 2>&1 | tee $R/gpurun_out/budget_sweep_10m.log
timeout -k 10 300 python -u tools/big_budget_sweep.py 500 16 0 64 128 256 512 1024 2>&1 | tee $R/gpurun_out/budget_sweep_1m.log
MIRT_LIBMIRT=$R/variants/stretch/libmirt.so timeout -k 10 300 python -u bench.py --scene synthetic_bvh:1581 --spp 8 --steps 2 --warmup 1 > $R/gpurun_out/stretch_10m.json
python -c "import json; d=json.load(open('$R/gpurun_out/stretch_10m.json')); print('stretches 10M', d['value'])"
MIRT_LIBMIRT=$R/variants/stretch/libmirt.so timeout -k 10 300 python -u bench.py --scene synthetic_bvh:500 --spp 16 --steps 2 --warmup 1 > $R/gpurun_out/stretch_1m.json
python -c "import json; d=json.load(open('$R/gpurun_out/stretch_1m.json')); print('stretches 1M', d['value'])"
