#!/bin/bash
# configs[2] end to end + kernel stats and SQ counters of the decode kernels (V2 in bench.py, V1 in coco_mot17_bench).
set -e
export TMPDIR=/tmp
O=$PWD/gpurun_out/r03d
rm -rf $O && mkdir -p $O
R=$PWD
timeout -k 10 300 python3 -u tools/coco_mot17_bench.py --json $O/coco_mot17.json > $O/coco_mot17.log 2>&1 || { tail -20 $O/coco_mot17.log; exit 1; }
cat $O/coco_mot17.log | cut -c1-400
timeout -k 10 200 python3 -u tools/decode_bench.py > $O/decode_bench.txt 2>&1
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/v1stats -- python3 $R/tools/coco_mot17_bench.py --steps 3 --people 60 > $O/v1stats.log 2>&1
C="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
BENCH_NO_H2D_LEG=1 timeout -s KILL 300 rocprofv3 --pmc $C --output-format csv -d $O/pmc_v2 -- python3 $R/bench.py --streams 1 --steps 2 --warmup 1 --cpu-frames 0 --sustained-steps 0 > $O/pmc_v2.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc $C --output-format csv -d $O/pmc_v1 -- python3 $R/tools/coco_mot17_bench.py --steps 2 --people 60 > $O/pmc_v1.log 2>&1
cd $R
python3 tools/pmc_table.py $O/pmc_v2 'nms_|paf_|assemble|preprocess' > $O/pmc_v2.md
python3 tools/pmc_table.py $O/pmc_v1 'v1_|preprocess' > $O/pmc_v1.md
cat $O/pmc_v2.md $O/pmc_v1.md
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete; find $O -name "*agent_info.csv" -delete
