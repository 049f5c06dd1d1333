# tree quality of the two builders on a large scene: usage gpu_builder_cmp.sh HALF SPP
for b in device host; do
  echo "builder $b"
  MIRT_VERBOSE= MIRT_BVH_BUILDER=$b timeout -k 10 900 python -u tools/big_env_sweep.py $1 $2 MIRT_STEP_BUDGET 256 2>&1 | grep -v "^mirt:" || exit 1
done
