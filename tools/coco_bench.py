#!/usr/bin/env python3
"""Forward timing of the COCO-18 graph (BASELINE configs[2]: cv2.dnn flavour, 368x656 net input)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pose-estimation_amd"))
from posehip import _ffi, caffe_io, graph, synth  # noqa: E402
from posehip.engine import Engine  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
H, W = 368, 656
net = caffe_io.parse_prototxt(caffe_io.coco_prototxt(H, W))
plan = graph.build_plan(net, graph.random_weights(net, 0))
eng = Engine(0)
eng.build(plan, B)
eng.preprocess(synth.synth_frames(1, B, H, W), _ffi.PH_PRE_CV2)
for _ in range(3):
    ms = eng.forward_timed(B)
fl = eng.op_flops() * B
names = eng.op_kernels()
by = {}
for n, t, f in zip(names, ms, fl):
    a = by.setdefault(n, [0.0, 0.0, 0])
    a[0] += t; a[1] += f; a[2] += 1
for n, (t, f, c) in sorted(by.items(), key=lambda kv: -kv[1][0]):
    print(f"{n:48s} {c:3d} launches {t:8.3f} ms {f / max(t, 1e-9) / 1e9:8.1f} TFLOP/s")
print(f"COCO-18 368x656 B={B}: forward {ms.sum():.3f} ms = {B / ms.sum() * 1e3:.0f} frames/s, "
      f"{plan.flops_per_frame() / 1e9:.1f} GFLOP/frame, {fl.sum() / ms.sum() / 1e9:.0f} TFLOP/s")
