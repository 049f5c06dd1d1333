set -e
R=$GRAFT_REPO_ROOT
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread -k "not bench" > $R/gpurun_out/wide_tests.log 2>&1 || { tail -30 $R/gpurun_out/wide_tests.log; exit 1; }
tail -2 $R/gpurun_out/wide_tests.log
for v in binary new; do
  if [ $v = binary ]; then export MIRT_LIBMIRT=$R/variants/binary/libmirt.so; else unset MIRT_LIBMIRT; fi
  echo "== $v"
  timeout -k 10 300 python -u tools/big_budget_sweep.py 40 256 0
  timeout -k 10 300 python -u tools/big_budget_sweep.py 500 32 128 256 512
  timeout -k 10 600 python -u tools/big_budget_sweep.py 1581 48 256 512 1024 2048
done 2>&1 | tee $R/gpurun_out/wide_ab.log
