timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/pytest_gpu.log 2>&1; rc=$?; echo exit=$rc; tail -4 gpurun_out/pytest_gpu.log | cut -c1-300; [ $rc -eq 0 ] || exit 1
timeout -k 10 300 python -u bench.py --no-cpu-baseline > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err || { tail -3 gpurun_out/bench_default.err; exit 1; }
python -c "
import json; d=json.load(open('gpurun_out/bench_default.json')); print('bench', round(d['value'],1), d['ms_per_step'], d['config'], d['roofline']['issue']['frac'], d['roofline']['hbm_measured'])"
timeout -k 10 800 python -u tools/big_env_sweep.py 1581 64 POOL 8388608 33554432 || exit 1
