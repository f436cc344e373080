#!/bin/bash
# Host-frame leg under different stream / hardware-queue counts (1080p source frames).
set -e
export TMPDIR=/tmp
O=$PWD/gpurun_out/h2dv
rm -rf $O && mkdir -p $O
run() { # name, env..., -- args
  name=$1; shift
  timeout -k 10 200 env "$@" python3 bench.py --frame-size 1080x1920 --steps 100 --warmup 6 --cpu-frames 0 --sustained-steps 0 $EXTRA > $O/$name.json 2> $O/$name.err
  python3 -c "
import json,sys
d=json.loads(open('$O/$name.json').read().strip().splitlines()[-1])
print('$name', d['value'], d['ms_per_step'], d['h2d_inclusive']['value'], d['h2d_inclusive']['frac_of_resident'])"
}
EXTRA="--streams 3" run q4_s3 X=1
EXTRA="--streams 2" run q4_s2 X=1
EXTRA="--streams 3" run q8_s3 GPU_MAX_HW_QUEUES=8
EXTRA="--streams 4" run q8_s4 GPU_MAX_HW_QUEUES=8
EXTRA="--streams 2" run q8_s2 GPU_MAX_HW_QUEUES=8
EXTRA="--streams 3" run q4_s3_nosdma HSA_ENABLE_SDMA=0
