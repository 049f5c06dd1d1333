# instruction-cache and fetch counters of the render kernel (one pass)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/pmc_c
timeout -s KILL 60 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_INSTS_VALU -d $R/gpurun_out/pmc_c -o c --output-format csv -- python3 $R/tools/pmc_probe.py > $R/gpurun_out/pmc_c.log 2>&1 || { tail -5 $R/gpurun_out/pmc_c.log; exit 1; }
python3 $R/tools/pmc_summary.py $R/gpurun_out/pmc_c
