# the GPU test suite, as the round's end runs it; the log under gpurun_out/
mkdir -p gpurun_out
timeout -k 10 1100 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/suite.log 2>&1
rc=$?
tail -n 15 gpurun_out/suite.log
exit $rc
