# full default bench, then a kernel trace of a shorter run of the same command
set -e
R=$GRAFT_REPO_ROOT
timeout -k 10 500 python -u bench.py > $R/gpurun_out/bench_full.json 2> $R/gpurun_out/bench_full.err
python -c "
import json; d=json.load(open('$R/gpurun_out/bench_full.json')); print('full', round(d['value'],1), d['unit'], round(d['ms_per_step'],1), 'ms/step; cpu', d['cpu_baseline']['value'], 'roofline frac', d['roofline']['frac'])"
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof -o r --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/bench_prof.json 2> $R/gpurun_out/bench_prof.err
ls $R/gpurun_out/prof
head -5 $R/gpurun_out/prof/*kernel_stats.csv
