#!/bin/bash
# A/B of the fused 1x1 pair kernel (conv_pair.hip): headline bench with the per-op table, pair kernel off and on.
set -e
export TMPDIR=/tmp
O=$PWD/gpurun_out/pair_ab
rm -rf $O && mkdir -p $O
for f in 0 1 0 1; do
  POSEHIP_PAIR_KERNEL=$f timeout -k 10 300 python3 bench.py --cpu-frames 0 --per-op > $O/bench_$f.json 2> $O/bench_$f.err
  python3 - $f <<'PY'
import json, sys
f = sys.argv[1]
d = json.loads(open(f"gpurun_out/pair_ab/bench_{f}.json").read().strip().splitlines()[-1])
print("pair", f, d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["all_conv"]["conv_ms_per_step"])
PY
done
grep -n "Mconv6\|Mconv7\|pair" $O/bench_0.err | head -40
grep -n "Mconv6\|Mconv7\|pair" $O/bench_1.err | head -40
