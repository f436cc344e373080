#!/bin/bash
# Regenerates the raw material of profiles/rNN_* on one box (run from the repo root on the GPU box):
#   bash tools/refresh_profiles.sh        -> gpurun_out/prof/{bench.json,stats,stats3,pmc_fetch,pmc_write,...}
# Every GPU step has its own time limit and the script stops at the first failure.
set -e
export TMPDIR=/tmp
O=$PWD/gpurun_out/prof
rm -rf $O && mkdir -p $O
timeout -k 10 300 python3 __graft_entry__.py smoke > $O/smoke.log 2>&1 || { tail -5 $O/smoke.log; exit 1; }
timeout -k 10 400 python3 bench.py > $O/bench.json 2> $O/bench.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --streams 1 --steps 5 --warmup 2 --cpu-frames 0 --sustained-steps 0 > $O/stats.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats3 -- python3 bench.py --steps 6 --warmup 3 --cpu-frames 0 --sustained-steps 0 > $O/stats3.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 bench.py --streams 1 --steps 3 --warmup 1 --cpu-frames 0 --sustained-steps 0 > $O/pmc_fetch.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 bench.py --streams 1 --steps 3 --warmup 1 --cpu-frames 0 --sustained-steps 0 > $O/pmc_write.log 2>&1
python3 tools/pmc_traffic.py $O/pmc_fetch $O/pmc_write $O/stats $O/hbm_traffic.json > $O/traffic.log 2>&1
timeout -k 10 300 python3 -u tools/decode_bench.py > $O/decode_bench.txt 2>&1
timeout -k 10 120 python3 -u tools/overlap_probe.py > $O/overlap_probe.txt 2>&1
timeout -k 10 120 python3 -u tools/conv_bench.py > $O/conv_bench.txt 2>&1
timeout -k 10 200 python3 -u tools/latency_bench.py > $O/latency_bench.txt 2>&1
# keep the merge small: the per-kernel stats and the json, not the raw traces
find $O -name "*kernel_trace.csv" -delete
find $O -name "*counter_collection.csv" -delete
find $O -name "*agent_info.csv" -delete
ls -R $O | head -50
