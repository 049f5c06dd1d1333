# first GPU pass of the round: VALU issue probe, GPU tests, bench line
set -e
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cat /sys/fs/cgroup/cpu.max > $R/gpurun_out/cpu_max.txt 2>&1 || true
nproc >> $R/gpurun_out/cpu_max.txt; echo "OMP=$OMP_NUM_THREADS" >> $R/gpurun_out/cpu_max.txt
timeout -k 10 120 $R/tools/bin/valu_issue_probe > $R/gpurun_out/valu_issue_probe.jsonl
cat $R/gpurun_out/valu_issue_probe.jsonl | cut -c1-200
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 240 --timeout-method thread > $R/gpurun_out/gputests.log 2>&1 || { tail -30 $R/gpurun_out/gputests.log; exit 1; }
tail -3 $R/gpurun_out/gputests.log
timeout -k 10 400 python -u bench.py > $R/gpurun_out/bench_r2_a.json 2> $R/gpurun_out/bench_r2_a.err
cat $R/gpurun_out/bench_r2_a.json
