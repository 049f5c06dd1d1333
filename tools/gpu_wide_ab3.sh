set -e
R=$GRAFT_REPO_ROOT
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread -k "not bench" > $R/gpurun_out/wide_tests.log 2>&1 || { tail -30 $R/gpurun_out/wide_tests.log; exit 1; }
tail -2 $R/gpurun_out/wide_tests.log
( timeout -k 10 300 python -u tools/big_budget_sweep.py 40 256 0
  timeout -k 10 300 python -u tools/big_budget_sweep.py 500 32 128 256
  timeout -k 10 600 python -u tools/big_budget_sweep.py 1581 48 256 512 ) 2>&1 | tee $R/gpurun_out/wide_ab3.log
