# usage: gpu_sweep_lib.sh NAME VAR v1 v2 ...  -- bench variants/NAME (or the tree's build for NAME = tree) at 256 spp for each value of VAR
name=$1; var=$2; shift 2
lib=$PWD/variants/$name/libmirt.so; [ "$name" = tree ] && lib=
for v in "$@"; do
  env MIRT_LIBMIRT=$lib $var=$v timeout -k 10 200 python -u bench.py --steps 2 --warmup 1 --spp 256 --no-cpu-baseline > gpurun_out/sweep.json 2>gpurun_out/sweep.err || exit 1
  python -c "
import json; d=json.load(open('gpurun_out/sweep.json')); print('$name $var=$v', round(d['value'],1), round(d['ms_per_step'],2), {k: round(v,3) for k,v in d['roofline']['lanes_at_work'].items()})"
done
