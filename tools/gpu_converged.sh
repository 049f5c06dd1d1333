timeout -k 10 600 python -u -m pytest tests/test_gpu_converged.py -m gpu -x -q -s --timeout 300 --timeout-method thread 2>&1 | grep -E "dB|passed|failed|Error" | cut -c1-300 || exit 1
timeout -k 10 400 python -u bench.py > gpurun_out/bench_golden.json 2> gpurun_out/bench_golden.err || { tail -3 gpurun_out/bench_golden.err; exit 1; }
python -c "
import json; d=json.load(open('gpurun_out/bench_golden.json')); print('bench', round(d['value'],1), d['golden'], d['cpu_baseline'])"
