# collects SQ counters for extend/shade in two passes and prints per-kernel sums
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/pmc_a $R/gpurun_out/pmc_b
timeout -s KILL 60 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_INSTS_SALU -d $R/gpurun_out/pmc_a -o a --output-format csv -- python3 $R/tools/pmc_probe.py > $R/gpurun_out/pmc_a.log 2>&1 || { tail -5 $R/gpurun_out/pmc_a.log; exit 1; }
timeout -s KILL 60 rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CU_CYCLES -d $R/gpurun_out/pmc_b -o b --output-format csv -- python3 $R/tools/pmc_probe.py > $R/gpurun_out/pmc_b.log 2>&1 || { tail -5 $R/gpurun_out/pmc_b.log; exit 1; }
python3 $R/tools/pmc_summary.py $R/gpurun_out/pmc_a $R/gpurun_out/pmc_b
