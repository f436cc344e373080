#!/bin/bash
# Why does the host->device copy of the frames cost what it costs?  HIP API + memory-copy + kernel traces of the
# 1080p stream leg (no counters).
set -e
export TMPDIR=/tmp
O=$PWD/gpurun_out/h2d
rm -rf $O && mkdir -p $O
cd /tmp
timeout -k 10 300 rocprofv3 --hip-trace --memory-copy-trace --kernel-trace --stats --output-format csv -d $O/trace -- python3 $GRAFT_REPO_ROOT/bench.py --h2d --frame-size 1080x1920 --steps 20 --warmup 6 --cpu-frames 0 --sustained-steps 0 > $O/trace.log 2>&1
cd $GRAFT_REPO_ROOT
find $O -name "*kernel_trace.csv" -delete
find $O -name "*agent_info.csv" -delete
ls -laR $O | head -40
python3 - <<'PY'
import csv,glob
for f in glob.glob("gpurun_out/h2d/trace/**/*memory_copy_trace.csv", recursive=True):
    rows=list(csv.DictReader(open(f)))
    print(f, len(rows), rows[0].keys() if rows else None)
    big=[r for r in rows if int(r.get("Bytes", r.get("Size","0")) or 0) > 1e8]
    for r in big[:12]:
        d=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6
        print(r.get("Direction"), r.get("Bytes"), "ms", round(d,3), "GB/s", round(int(r["Bytes"])/d/1e6,1))
for f in glob.glob("gpurun_out/h2d/trace/**/*hip_api_stats.csv", recursive=True):
    print(open(f).read()[:3000])
PY
