# usage: gpu_suite_ab.sh NAME...  -- the GPU suite with the tree's build, then final_scene benched with each variants/NAME
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests -m gpu -q --timeout 300 --timeout-method thread > gpurun_out/suite.log 2>&1
rc=$?; tail -n 6 gpurun_out/suite.log | cut -c1-300; [ $rc -eq 0 ] || exit $rc
bash tools/gpu_ab.sh "$@"
