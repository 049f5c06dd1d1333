# Everything the round's figures come from, both scenes, raw outputs under gpurun_out/TAG
# (tools/make_counter_stamps.py turns them into profiles/*): bench lines, kernel stats of the same commands, SQ
# counters in two passes, FETCH_SIZE and WRITE_SIZE in passes of their own, work counts of the same frames.
# usage: bash tools/gpu_r3_profile.sh TAG
R=$GRAFT_REPO_ROOT
TAG=${1:-r3_a}
O=$R/gpurun_out/$TAG
rm -rf $O; mkdir -p $O
SQ_A="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_INSTS_SALU"
SQ_B="SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CU_CYCLES"
cd $R
timeout -k 10 500 python -u bench.py > $O/bench.json 2> $O/bench.err || { tail -3 $O/bench.err; exit 1; }
timeout -k 10 500 python -u bench.py --scene synthetic_bvh:1581 --spp 64 --steps 2 --warmup 1 > $O/big10m_bench.json 2> $O/big10m_bench.err || { tail -3 $O/big10m_bench.err; exit 1; }
timeout -k 10 500 python -u bench.py --scene synthetic_bvh:1581 --spp 512 --steps 1 --warmup 1 --no-cpu-baseline > $O/big10m_bench_512spp.json 2> $O/big10m_bench_512spp.err || { tail -3 $O/big10m_bench_512spp.err; exit 1; }
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/prof -o r --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_under_rocprofv3.json 2> $O/bench_prof.err || { tail -3 $O/bench_prof.err; exit 1; }
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/big_prof -o r --output-format csv -- python3 $R/bench.py --scene synthetic_bvh:1581 --spp 64 --steps 2 --warmup 1 --no-cpu-baseline > $O/big10m_bench_under_rocprofv3.json 2> $O/big_prof.err || { tail -3 $O/big_prof.err; exit 1; }
for pass in a b; do
  [ $pass = a ] && C=$SQ_A || C=$SQ_B
  timeout -s KILL 120 rocprofv3 --pmc $C -d $O/pmc_$pass -o $pass --output-format csv -- python3 $R/tools/pmc_probe.py final_scene 32 > $O/pmc_$pass.log 2>&1 || { tail -3 $O/pmc_$pass.log; exit 1; }
  timeout -s KILL 300 rocprofv3 --pmc $C -d $O/big_pmc_$pass -o $pass --output-format csv -- python3 $R/tools/big_bench_probe.py 1581 64 > $O/big_pmc_$pass.log 2>&1 || { tail -3 $O/big_pmc_$pass.log; exit 1; }
done
python3 $R/tools/pmc_summary.py $O/pmc_a $O/pmc_b > $O/pmc_sq_counters.txt && tail -1 $O/pmc_a.log >> $O/pmc_sq_counters.txt
python3 $R/tools/pmc_summary.py $O/big_pmc_a $O/big_pmc_b > $O/big10m_pmc_sq_counters.txt && tail -1 $O/big_pmc_a.log >> $O/big10m_pmc_sq_counters.txt
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 120 rocprofv3 --pmc $c -d $O/pmc_$c -o c --output-format csv -- python3 $R/tools/pmc_probe.py final_scene 1024 > $O/pmc_$c.log 2>&1 || { tail -3 $O/pmc_$c.log; exit 1; }
  timeout -s KILL 300 rocprofv3 --pmc $c -d $O/big_pmc_$c -o c --output-format csv -- python3 $R/tools/big_bench_probe.py 1581 64 > $O/big_pmc_$c.log 2>&1 || { tail -3 $O/big_pmc_$c.log; exit 1; }
done
grep -h k_render $O/pmc_FETCH_SIZE/*counter_collection.csv $O/pmc_WRITE_SIZE/*counter_collection.csv | cut -c1-400 > $O/pmc_fetch_write_rows_1024spp.csv
grep -h k_render $O/big_pmc_FETCH_SIZE/*counter_collection.csv $O/big_pmc_WRITE_SIZE/*counter_collection.csv | cut -c1-400 > $O/big10m_pmc_fetch_write_rows.csv
cd $R
timeout -k 10 200 python3 tools/phase_probe.py final_scene 1024 > $O/phase_1024.log 2>&1 || { tail -3 $O/phase_1024.log; exit 1; }
timeout -k 10 200 python3 tools/phase_probe.py final_scene 32 > $O/phase_32.log 2>&1 || { tail -3 $O/phase_32.log; exit 1; }
timeout -k 10 300 python3 tools/big_bench_probe.py 1581 64 count > $O/big10m_count.log 2>&1 || { tail -3 $O/big10m_count.log; exit 1; }
cp $O/prof/*kernel_stats.csv $O/kernel_stats.csv; cp $O/big_prof/*kernel_stats.csv $O/big10m_kernel_stats.csv
rm -rf $O/prof $O/big_prof $O/pmc_a $O/pmc_b $O/big_pmc_a $O/big_pmc_b $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/big_pmc_FETCH_SIZE $O/big_pmc_WRITE_SIZE
ls $O
python3 -c "
import json
for f in ('bench.json','big10m_bench.json','big10m_bench_512spp.json'):
    d=json.load(open('$O/'+f)); print(f, round(d['value'],1), d['unit'], round(d['ms_per_step'],2), 'ms', d['roofline']['bound'], d['roofline']['frac'])"
