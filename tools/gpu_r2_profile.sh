# final_scene at the benchmark's size: bench line, kernel stats of the same command, SQ counters, HBM-side traffic
# usage: bash tools/gpu_r2_profile.sh TAG
set -e
R=$GRAFT_REPO_ROOT
TAG=${1:-r2_c}
O=$R/gpurun_out/$TAG
rm -rf $O; mkdir -p $O
cd $R
timeout -k 10 500 python -u bench.py > $O/bench.json 2> $O/bench.err
python -c "
import json; d=json.load(open('$O/bench.json')); print('bench', round(d['value'],1), d['unit'], round(d['ms_per_step'],1), 'ms/step; cpu', d['cpu_baseline']['value'], d['cpu_baseline']['cores'], 'roofline', d['roofline']['bound'], d['roofline']['frac'])"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/prof -o r --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_prof.json 2> $O/bench_prof.err
head -3 $O/prof/*kernel_stats.csv
timeout -s KILL 90 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_INSTS_SALU -d $O/pmc_a -o a --output-format csv -- python3 $R/tools/pmc_probe.py final_scene 32 > $O/pmc_a.log 2>&1
timeout -s KILL 90 rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CU_CYCLES -d $O/pmc_b -o b --output-format csv -- python3 $R/tools/pmc_probe.py final_scene 32 > $O/pmc_b.log 2>&1
python3 $R/tools/pmc_summary.py $O/pmc_a $O/pmc_b > $O/pmc_sq_counters.txt
tail -1 $O/pmc_a.log >> $O/pmc_sq_counters.txt
cat $O/pmc_sq_counters.txt
timeout -s KILL 90 rocprofv3 --pmc FETCH_SIZE -d $O/pmc_f -o f --output-format csv -- python3 $R/tools/pmc_probe.py final_scene 64 > $O/pmc_f.log 2>&1
timeout -s KILL 90 rocprofv3 --pmc WRITE_SIZE -d $O/pmc_w -o w --output-format csv -- python3 $R/tools/pmc_probe.py final_scene 64 > $O/pmc_w.log 2>&1
grep -h k_render $O/pmc_f/*counter_collection.csv $O/pmc_w/*counter_collection.csv | cut -c1-400 > $O/pmc_fetch_write_rows.csv
cd $R
timeout -k 10 120 python3 tools/phase_probe.py final_scene 32 > $O/phase_32.log 2>&1 || true
timeout -k 10 120 python3 tools/phase_probe.py final_scene 64 > $O/phase_64.log 2>&1 || true
tail -n 2 $O/phase_32.log; tail -n 2 $O/phase_64.log
