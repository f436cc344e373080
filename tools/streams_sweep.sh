#!/bin/bash
# Batches in flight (engines / HIP streams) against frames/s, now that every stream has a hardware queue of its own.
set -e
O=$PWD/gpurun_out/streams
rm -rf $O && mkdir -p $O
for s in 3 4 5 3 4 6; do
  timeout -k 10 300 python3 bench.py --cpu-frames 0 --streams $s > $O/b_$s.json 2> $O/b_$s.err
  python3 - $s <<'PY'
import json, sys
s = sys.argv[1]
d = json.loads(open(f"gpurun_out/streams/b_{s}.json").read().strip().splitlines()[-1])
print("streams", s, "frames/s", d["value"], "ms/step", d["ms_per_step"], "h2d", d.get("h2d_inclusive", {}).get("frac_of_resident"), "hw_queues", d["config"].get("hw_queues"))
PY
done
