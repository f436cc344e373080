#!/usr/bin/env python3
"""The stem op alone (conv1_1 + conv1_2 + pool at 368x656, batch 32): device ms of the persistent kernel and, with
POSEHIP_STEM_KERNEL=0, of the halo kernel's stem form."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "pose-estimation_amd"))
import numpy as np
from posehip import _ffi, caffe_io, graph, synth
from posehip.engine import Engine
B, H, W = 32, 368, 656
txt = f'input: "image"\ninput_dim: 1\ninput_dim: 3\ninput_dim: {H}\ninput_dim: {W}\n'
txt += caffe_io._conv("c11", "image", 64, 3) + caffe_io._act("ReLU", "r11", "c11")
txt += caffe_io._conv("c12", "c11", 64, 3) + caffe_io._act("ReLU", "r12", "c12")
txt += 'layer { name: "p1" type: "Pooling" bottom: "c12" top: "p1" pooling_param { pool: MAX kernel_size: 2 stride: 2 } }\n'
txt += caffe_io._conv("out", "p1", 8, 1)
net = caffe_io.parse_prototxt(txt)
plan = graph.build_plan(net, graph.random_weights(net, 0, out_gain=1.0))
frames = synth.synth_frames(0, B, H, W)
for flag in ("1", "0"):
    os.environ["POSEHIP_STEM_KERNEL"] = flag
    eng = Engine(0)
    eng.build(plan, B)
    eng.preprocess(frames, _ffi.PH_PRE_CV2)
    for _ in range(30):
        eng.forward_timed(B)
    ms = np.median([eng.forward_timed(B)[1] for _ in range(30)])
    print(eng.op_kernels()[1], f"{ms:.4f} ms", flush=True)
    eng.close()
