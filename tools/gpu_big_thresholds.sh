# Walk thresholds for scenes in global memory, at three sizes.   usage: gpu_big_thresholds.sh TAG
set -e
O=gpurun_out/$1; mkdir -p $O
for half_spp in "1581 64" "500 32" "40 256"; do
  set -- $half_spp
  for rn in "32 16" "48 32" "56 32" "48 40" "56 40" "56 48"; do
    set -- $half_spp $rn
    echo "refill $3 node $4"
    MIRT_REFILL_MIN=$3 MIRT_NODE_MIN=$4 timeout -k 10 300 python -u tools/big_env_sweep.py $1 $2 MIRT_STEP_BUDGET 256 | tail -1
  done
done | tee $O/thresholds.txt
