import os, sys
import numpy as np, torch
sys.path.insert(0, "pose-estimation_amd")
from posehip import _ffi, caffe_io, graph, synth
from posehip.engine import Engine
H, W, B = 368, 656, 32
net = caffe_io.parse_prototxt(caffe_io.body25_prototxt(H, W))
plan = graph.build_plan(net, graph.random_weights(net, 0))
ts = torch.cuda.Stream()
eng = Engine(0, ts.cuda_stream); eng.build(plan, B)
for (h0, w0) in ((368, 656), (1080, 1920)):
    f = torch.from_numpy(synth.synth_frames(3, B, h0, w0)).cuda()
    for mode, name in ((_ffi.PH_PRE_CV2, "cv2"), (_ffi.PH_PRE_NATIVE, "native")):
        with torch.cuda.stream(ts):
            for _ in range(3): eng.preprocess(f, mode)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(ts)
            for _ in range(10): eng.preprocess(f, mode)
            e1.record(ts)
        ts.synchronize()
        ms = e0.elapsed_time(e1) / 10
        rb = B * h0 * w0 * 3; wb = B * (H + 2) * (W + 2) * 8 * 2
        print(f"preprocess {name:6s} {w0}x{h0} -> {W}x{H} B={B}: {ms:.3f} ms  ({(rb + wb) / ms / 1e9:.2f} TB/s of read+write)")
