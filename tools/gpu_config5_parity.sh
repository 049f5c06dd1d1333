# config 5 at its own size: ids and t bits of the 10 M-sphere scene against the oracle's linear World::hit scan, then
# 40 000 more rays against the culling container (itself checked against the scan on the first set)
mkdir -p gpurun_out
timeout -k 10 1000 python -u tools/big_parity_probe.py 1581 32 linear 40000 > gpurun_out/big_parity_1581.log 2>&1
rc=$?
tail -n 12 gpurun_out/big_parity_1581.log
exit $rc
