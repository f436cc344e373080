#!/usr/bin/env python3
"""Single-frame latency (BASELINE configs[0]: BODY_25, one 368x368 frame; and 368x656): preprocess + forward +
V2 decode + results on the host, one frame at a time, synchronised per frame."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pose-estimation_amd"))
from posehip import _ffi, caffe_io, graph, synth  # noqa: E402
from posehip.engine import Engine  # noqa: E402
from posehip.tables import BODY_25  # noqa: E402

for H, W in ((368, 368), (368, 656)):
    net = caffe_io.parse_prototxt(caffe_io.body25_prototxt(H, W))
    plan = graph.build_plan(net, graph.random_weights(net, 0))
    eng = Engine(0)
    eng.build(plan, 1)
    frame = synth.synth_frames(5, 1, H, W)
    out = synth.synth_net_output(77, 3, BODY_25, H // 8, W // 8)[None]
    for inject in (False, True):
        ts = []
        for i in range(60):
            t0 = time.perf_counter()
            eng.preprocess(frame, _ffi.PH_PRE_NATIVE)
            eng.forward(1)
            if inject:          # realistic decode input (3 people) instead of the random net's noise
                eng.set_net_out(out)
            res, _ = eng.decode_v2(BODY_25, 1, H, W, 1.0, 128)
            ts.append(time.perf_counter() - t0)
        ts = np.array(ts[10:]) * 1e3
        ms = eng.forward_timed(1)
        print(f"{W}x{H} B=1 {'3-people maps' if inject else 'noise maps'}: median {np.median(ts):.3f} ms/frame "
              f"(p90 {np.percentile(ts, 90):.3f}); forward kernels alone {ms.sum():.3f} ms over {len(ms)} ops")
    eng.close()
