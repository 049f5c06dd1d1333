# HBM-side traffic of the render kernel: FETCH_SIZE and WRITE_SIZE in separate counter passes (64 spp frame)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/pmc_f $R/gpurun_out/pmc_w
timeout -s KILL 60 rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/pmc_f -o f --output-format csv -- python3 $R/tools/pmc_probe.py final_scene 64 > $R/gpurun_out/pmc_f.log 2>&1 || { tail -5 $R/gpurun_out/pmc_f.log; exit 1; }
timeout -s KILL 60 rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/pmc_w -o w --output-format csv -- python3 $R/tools/pmc_probe.py final_scene 64 > $R/gpurun_out/pmc_w.log 2>&1 || { tail -5 $R/gpurun_out/pmc_w.log; exit 1; }
tail -1 $R/gpurun_out/pmc_f.log; tail -1 $R/gpurun_out/pmc_w.log
grep k_render $R/gpurun_out/pmc_f/*counter_collection.csv | cut -c1-400
grep k_render $R/gpurun_out/pmc_w/*counter_collection.csv | cut -c1-400
