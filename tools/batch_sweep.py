import os, sys
import numpy as np
sys.path.insert(0, "pose-estimation_amd")
from posehip import _ffi, caffe_io, graph, synth
from posehip.engine import Engine
H, W = 368, 656
net = caffe_io.parse_prototxt(caffe_io.body25_prototxt(H, W))
plan = graph.build_plan(net, graph.random_weights(net, 0))
for B in (1, 2, 4, 8, 16, 32, 64, 96):   # 64 / 96: what one launch per layer over two / three batches would do
    eng = Engine(0); eng.build(plan, B)
    eng.preprocess(synth.synth_frames(1, B, H, W), _ffi.PH_PRE_NATIVE)
    for _ in range(3): ms = eng.forward_timed(B)
    print(f"BODY_25 B={B:2d}: forward {ms.sum():.3f} ms = {B / ms.sum() * 1e3:.0f} frames/s")
    eng.close()
