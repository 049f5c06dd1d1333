# SQ counters of the global-memory walk (k_render<false,false>) on config 5's scene, one render per pass.
# usage: bash tools/gpu_big_sq.sh [half=1581] [spp=64] [tag=r3_big10m]
R=$GRAFT_REPO_ROOT
HALF=${1:-1581}; SPP=${2:-64}; TAG=${3:-r3_big10m}
O=$R/gpurun_out/$TAG
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_INSTS_SALU -d $O/pmc_a -o a --output-format csv -- python3 $R/tools/big_bench_probe.py $HALF $SPP > $O/pmc_a.log 2>&1 || { tail -3 $O/pmc_a.log; exit 1; }
timeout -s KILL 300 rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CU_CYCLES -d $O/pmc_b -o b --output-format csv -- python3 $R/tools/big_bench_probe.py $HALF $SPP > $O/pmc_b.log 2>&1 || { tail -3 $O/pmc_b.log; exit 1; }
python3 $R/tools/pmc_summary.py $O/pmc_a $O/pmc_b > $O/pmc_sq_counters.txt
tail -1 $O/pmc_a.log >> $O/pmc_sq_counters.txt
cat $O/pmc_sq_counters.txt
cd $R
timeout -k 10 300 python3 tools/big_bench_probe.py $HALF $SPP count > $O/count.log 2>&1 || exit 1
tail -1 $O/count.log
rm -rf $O/pmc_a $O/pmc_b
