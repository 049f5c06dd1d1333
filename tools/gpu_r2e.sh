# sweeps of the walk's thresholds and the pool with wave-owned regions, then SQ counters of a 32-spp frame
bash tools/gpu_sweep.sh MIRT_REFILL_MIN 16 24 32 40 48 || exit 1
bash tools/gpu_sweep.sh MIRT_NODE_MIN 8 16 24 32 || exit 1
bash tools/gpu_pool_sweep.sh 2097152 8388608 33554432 || exit 1
bash tools/gpu_pmc.sh > gpurun_out/pmc_sq.txt 2>&1 || { tail -5 gpurun_out/pmc_sq.txt; exit 1; }
cat gpurun_out/pmc_sq.txt
timeout -k 10 200 python -u tools/phase_probe.py final_scene 32
