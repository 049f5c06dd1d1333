# the overlap test with the library as built, the whole GPU suite with the max-ilp build of the same source,
# then both builds benched back to back
mkdir -p gpurun_out
timeout -k 10 300 python -u -m pytest tests/test_zz_no_alloc.py -m gpu -x -q --timeout 200 --timeout-method thread > gpurun_out/no_alloc.log 2>&1
rc=$?; tail -n 3 gpurun_out/no_alloc.log; [ $rc -eq 0 ] || exit $rc
MIRT_LIBMIRT=$PWD/variants/maxilp/libmirt.so timeout -k 10 500 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/suite_maxilp.log 2>&1
rc=$?; tail -n 5 gpurun_out/suite_maxilp.log; [ $rc -eq 0 ] || exit $rc
bash tools/gpu_ab.sh base maxilp
