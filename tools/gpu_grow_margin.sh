# usage: gpu_grow_margin.sh SELECT NAME...  -- the grazing-ray tests whose id matches SELECT ("4500": four-child trees
# walked from global memory; "400": scenes staged in LDS) with the tree's build (must pass), then with each
# variants/NAME (builds with smaller growth constants: where do wrong answers begin?)
sel="test_rays_that_graze_spheres_match_the_exhaustive_scan and $1"; shift
timeout -k 10 300 python -u -m pytest tests/test_gpu_fuzz.py -m gpu -x -q -k "$sel" --timeout 200 --timeout-method thread 2>&1 | tail -2 || exit 1
for v in "$@"; do
  echo "== $v"
  MIRT_LIBMIRT=$PWD/variants/$v/libmirt.so timeout -k 10 300 python -u -m pytest tests/test_gpu_fuzz.py -m gpu -q -k "$sel" --timeout 200 --timeout-method thread 2>&1 | grep -E "^E +AssertionError|passed|failed" | cut -c1-200
done
exit 0
