# HBM-side traffic of the benchmark's own frame (1024 spp, pool sized by the library): FETCH_SIZE and WRITE_SIZE in separate passes
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/pmc_f $R/gpurun_out/pmc_w
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/pmc_f -o f --output-format csv -- python3 $R/tools/pmc_probe.py final_scene 1024 > $R/gpurun_out/pmc_f.log 2>&1 || { tail -5 $R/gpurun_out/pmc_f.log; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/pmc_w -o w --output-format csv -- python3 $R/tools/pmc_probe.py final_scene 1024 > $R/gpurun_out/pmc_w.log 2>&1 || { tail -5 $R/gpurun_out/pmc_w.log; exit 1; }
grep -h "final_scene" $R/gpurun_out/pmc_f.log $R/gpurun_out/pmc_w.log | grep rounds
grep -h k_render $R/gpurun_out/pmc_f/*counter_collection.csv $R/gpurun_out/pmc_w/*counter_collection.csv | cut -c1-400
cd $R && timeout -k 10 200 python3 tools/phase_probe.py final_scene 1024 | head -3
