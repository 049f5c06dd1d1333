# usage: gpu_grow_margin.sh NAME...  -- the grazing-ray tests on four-child trees with the tree's build (must pass), then
# with each variants/NAME (builds with a smaller BOX_GROW_ANY: where do wrong answers begin?)
sel='test_rays_that_graze_spheres_match_the_exhaustive_scan and 4500'
timeout -k 10 300 python -u -m pytest tests/test_gpu_fuzz.py -m gpu -x -q -k "$sel" --timeout 200 --timeout-method thread 2>&1 | tail -2 || exit 1
for v in "$@"; do
  echo "== $v"
  MIRT_LIBMIRT=$PWD/variants/$v/libmirt.so timeout -k 10 300 python -u -m pytest tests/test_gpu_fuzz.py -m gpu -q -k "$sel" --timeout 200 --timeout-method thread 2>&1 | grep -E "^E +AssertionError|passed|failed" | cut -c1-200
done
exit 0
