# Config 5's scene (10 M spheres) on one GPU: counter calibration, a bench line, kernel stats, HBM-side traffic.
# usage: bash tools/gpu_big_profile.sh [half=1581] [spp=8] [tag=r2_big]
set -e
R=$GRAFT_REPO_ROOT
HALF=${1:-1581}; SPP=${2:-64}; TAG=${3:-r2_big}
O=$R/gpurun_out/$TAG
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE -d $O/cal_f -o f --output-format csv -- $R/tools/bin/fetch_calibrate > $O/cal_f.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE -d $O/cal_w -o w --output-format csv -- $R/tools/bin/fetch_calibrate > $O/cal_w.log 2>&1
tail -1 $O/cal_f.log
grep -h -E "k_gather64|k_stream64|k_stream16" $O/cal_f/*counter_collection.csv $O/cal_w/*counter_collection.csv | cut -c1-300 > $O/calibration_rows.csv
cat $O/calibration_rows.csv | awk -F, '{print $(NF-1), $NF, $0}' | cut -c1-200
cd $R
timeout -k 10 500 python -u bench.py --scene synthetic_bvh:$HALF --spp $SPP --steps 2 --warmup 1 > $O/bench.json 2> $O/bench.err
cat $O/bench.json
cd /tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/prof -o r --output-format csv -- python3 $R/bench.py --scene synthetic_bvh:$HALF --spp $SPP --steps 2 --warmup 1 > $O/bench_prof.json 2> $O/bench_prof.err
head -4 $O/prof/*kernel_stats.csv
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE -d $O/pmc_f -o f --output-format csv -- python3 $R/tools/big_bench_probe.py $HALF $SPP > $O/pmc_f.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE -d $O/pmc_w -o w --output-format csv -- python3 $R/tools/big_bench_probe.py $HALF $SPP > $O/pmc_w.log 2>&1
tail -1 $O/pmc_f.log
grep -h k_render $O/pmc_f/*counter_collection.csv $O/pmc_w/*counter_collection.csv | cut -c1-400 > $O/pmc_rows.csv
timeout -k 10 300 python3 $R/tools/big_bench_probe.py $HALF $SPP count > $O/count.log 2>&1
tail -1 $O/count.log
rm -rf $O/cal_f $O/cal_w
