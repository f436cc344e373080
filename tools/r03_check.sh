#!/bin/bash
# Round-3 first pass on one box: GPU tests, headline bench, the 1080p stream leg (resident vs host frames).
set -e
export TMPDIR=/tmp
O=$PWD/gpurun_out/r03a
rm -rf $O && mkdir -p $O
timeout -k 10 900 python3 -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -3 $O/pytest.log
timeout -k 10 400 python3 bench.py > $O/bench.json 2> $O/bench.err
timeout -k 10 400 python3 bench.py --frame-size 1080x1920 --cpu-frames 0 > $O/bench_1080.json 2> $O/bench_1080.err
python3 - <<'PY'
import json
for f in ("bench.json","bench_1080.json"):
    d=json.loads(open("gpurun_out/r03a/"+f).read().strip().splitlines()[-1])
    print(f, d["value"], d["ms_per_step"], d.get("h2d_inclusive"), d["roofline"]["frac"], d["roofline"]["all_conv"]["conv_ms_per_step"])
PY
