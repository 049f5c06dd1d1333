set -e
R=$GRAFT_REPO_ROOT
for v in binary new; do
  if [ $v = binary ]; then export MIRT_LIBMIRT=$R/variants/binary/libmirt.so; else unset MIRT_LIBMIRT; fi
  echo "== $v"
  timeout -k 10 600 python -u tools/big_budget_sweep.py 1581 48 512
done 2>&1 | tee $R/gpurun_out/wide_ab2.log
