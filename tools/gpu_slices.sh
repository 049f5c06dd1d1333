set -e
R=$GRAFT_REPO_ROOT
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread -k "not large_scene and not bench" > $R/gpurun_out/slices_tests.log 2>&1 || { tail -30 $R/gpurun_out/slices_tests.log; exit 1; }
tail -2 $R/gpurun_out/slices_tests.log
( timeout -k 10 300 python -u tools/time_slice_probe.py spheres_bvh 256
  timeout -k 10 300 python -u tools/time_slice_probe.py synthetic_bvh:40 64
  timeout -k 10 300 python -u tools/time_slice_probe.py synthetic_bvh:500 32 1 4 8
  timeout -k 10 600 python -u tools/time_slice_probe.py synthetic_bvh:1581 32 1 4 ) 2>&1 | tee $R/gpurun_out/time_slices.log
