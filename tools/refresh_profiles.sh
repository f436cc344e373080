#!/bin/bash
# Regenerates the raw material of profiles/rNN_* on one box (run from the repo root on the GPU box):
#   bash tools/refresh_profiles.sh        -> gpurun_out/prof/{bench.json,stats,stats3,pmc_fetch,pmc_write,...}
# Every GPU step has its own time limit and the script stops at the first failure.
set -e
export TMPDIR=/tmp
O=$PWD/gpurun_out/prof
R=$PWD
rm -rf $O && mkdir -p $O
timeout -k 10 300 python3 __graft_entry__.py smoke > $O/smoke.log 2>&1 || { tail -5 $O/smoke.log; exit 1; }
timeout -k 10 400 python3 bench.py > $O/bench.json 2> $O/bench.err
timeout -k 10 400 python3 bench.py --frame-size 1080x1920 --cpu-frames 0 > $O/bench_1080p.json 2> $O/bench_1080p.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --streams 1 --steps 5 --warmup 2 --cpu-frames 0 --sustained-steps 0 > $O/stats.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats3 -- python3 bench.py --steps 6 --warmup 3 --cpu-frames 0 --sustained-steps 0 > $O/stats3.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 bench.py --streams 1 --steps 3 --warmup 1 --cpu-frames 0 --sustained-steps 0 > $O/pmc_fetch.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 bench.py --streams 1 --steps 3 --warmup 1 --cpu-frames 0 --sustained-steps 0 > $O/pmc_write.log 2>&1
python3 tools/pmc_traffic.py $O/pmc_fetch $O/pmc_write $O/stats $O/hbm_traffic.json > $O/traffic.log 2>&1
# SQ counters: the conv kernels one launch at a time (conv_bench), the decode kernels in the bench step (V2) and in
# the configs[2] step (V1)
C1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
C2="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
cd /tmp
for s in conv4_2 conv3_2 m384_128 m128 m288_96 m96 m6_512 m7_52 pair512 stem; do   # one shape per run: a launch on its own
  timeout -s KILL 120 rocprofv3 --pmc $C1 --output-format csv -d $O/pmc_conv_$s -- python3 $R/tools/conv_bench.py --reps 2 --shapes $s > $O/pmc_conv_$s.log 2>&1
done
BENCH_NO_H2D_LEG=1 timeout -s KILL 300 rocprofv3 --pmc $C2 --output-format csv -d $O/pmc_v2 -- python3 $R/bench.py --streams 1 --steps 2 --warmup 1 --cpu-frames 0 --sustained-steps 0 > $O/pmc_v2.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc $C2 --output-format csv -d $O/pmc_v1 -- python3 $R/tools/coco_mot17_bench.py --steps 2 --people 60 > $O/pmc_v1.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/v1stats -- python3 $R/tools/coco_mot17_bench.py --steps 3 --people 60 > $O/v1stats.log 2>&1
cd $R
: > $O/pmc_conv.md
for s in conv4_2 conv3_2 m384_128 m128 m288_96 m96 m6_512 m7_52 pair512 stem; do
  echo "shape $s" >> $O/pmc_conv.md
  python3 tools/pmc_table.py $O/pmc_conv_$s 'conv3x3|conv_stem|conv1x1_pair' >> $O/pmc_conv.md
done
python3 tools/pmc_table.py $O/pmc_v2 'nms_|paf_|assemble|preprocess' > $O/pmc_v2.md
python3 tools/pmc_table.py $O/pmc_v1 'v1_|preprocess' > $O/pmc_v1.md
timeout -k 10 300 python3 -u tools/coco_mot17_bench.py --json $O/coco_mot17.json > $O/coco_mot17.log 2>&1
timeout -k 10 300 python3 -u tools/decode_bench.py > $O/decode_bench.txt 2>&1
timeout -k 10 120 python3 -u tools/overlap_probe.py > $O/overlap_probe.txt 2>&1
timeout -k 10 120 python3 -u tools/conv_bench.py > $O/conv_bench.txt 2>&1
timeout -k 10 120 python3 -u tools/stem_ab.py > $O/stem_ab.txt 2>&1
timeout -k 10 200 python3 -u tools/latency_bench.py > $O/latency_bench.txt 2>&1
timeout -k 10 200 python3 -u tools/batch_sweep.py > $O/batch_sweep.txt 2>&1
# keep the merge small: the per-kernel stats and the json, not the raw traces
find $O -name "*kernel_trace.csv" -delete
find $O -name "*counter_collection.csv" -delete
find $O -name "*agent_info.csv" -delete
ls -R $O | head -60
