# The bench lines of tools/gpu_r3_profile.sh TAG once more, after tools/make_counter_stamps.py TAG has stamped the
# counter files with this tree's kernel hash: the lines then carry counters of their own build (`stale` false).
# usage: bash tools/gpu_r3_lines.sh TAG   (then: python tools/make_counter_stamps.py TAG lines)
R=$GRAFT_REPO_ROOT
TAG=${1:-r3_a}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
timeout -k 10 500 python -u bench.py > $O/bench.json 2> $O/bench.err || { tail -3 $O/bench.err; exit 1; }
timeout -k 10 500 python -u bench.py --scene synthetic_bvh:1581 --spp 64 --steps 2 --warmup 1 > $O/big10m_bench.json 2> $O/big10m_bench.err || { tail -3 $O/big10m_bench.err; exit 1; }
timeout -k 10 500 python -u bench.py --scene synthetic_bvh:1581 --spp 512 --steps 1 --warmup 1 --no-cpu-baseline > $O/big10m_bench_512spp.json 2> $O/big10m_bench_512spp.err || { tail -3 $O/big10m_bench_512spp.err; exit 1; }
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/prof -o r --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_under_rocprofv3.json 2> $O/bench_prof.err || { tail -3 $O/bench_prof.err; exit 1; }
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/big_prof -o r --output-format csv -- python3 $R/bench.py --scene synthetic_bvh:1581 --spp 64 --steps 2 --warmup 1 --no-cpu-baseline > $O/big10m_bench_under_rocprofv3.json 2> $O/big_prof.err || { tail -3 $O/big_prof.err; exit 1; }
cp $O/prof/*kernel_stats.csv $O/kernel_stats.csv; cp $O/big_prof/*kernel_stats.csv $O/big10m_kernel_stats.csv
rm -rf $O/prof $O/big_prof
python3 -c "
import json
for f in ('bench.json','big10m_bench.json','big10m_bench_512spp.json','bench_under_rocprofv3.json','big10m_bench_under_rocprofv3.json'):
    d=json.load(open('$O/'+f)); r=d['roofline']; print(f, round(d['value'],1), d['unit'], round(d['ms_per_step'],2), 'ms', r['bound'], round(r['frac'],4), 'stale' if (r.get('traffic_source') or {}).get('stale') or (r.get('issue') or {}).get('stale') else 'counters of this build')"
