#!/usr/bin/env python3
"""Diagnostic build only (-DPH_LIN_STAMP, see tools/conv_stamp.py): cycle stamps of the stem kernel
(conv1_1 evaluated inside conv1_2, fused 2x2 pool) on the 368x656 batch-32 shape."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "pose-estimation_amd"))
import numpy as np
from posehip import _ffi, caffe_io, graph, synth
from posehip.engine import Engine
eng = Engine(0); B = 32
H, W = 368, 656
txt = f'input: "image"\ninput_dim: 1\ninput_dim: 3\ninput_dim: {H}\ninput_dim: {W}\n'
txt += caffe_io._conv("c11", "image", 64, 3) + caffe_io._act("ReLU", "r11", "c11")
txt += caffe_io._conv("c12", "c11", 64, 3) + caffe_io._act("ReLU", "r12", "c12")
txt += 'layer { name: "p1" type: "Pooling" bottom: "c12" top: "p1" pooling_param { pool: MAX kernel_size: 2 stride: 2 } }\n'
txt += caffe_io._conv("out", "p1", 8, 1)
net = caffe_io.parse_prototxt(txt)
plan = graph.build_plan(net, graph.random_weights(net, 0, out_gain=1.0))
eng.build(plan, B)
print(eng.op_kernels())
eng.preprocess(synth.synth_frames(0, B, H, W), _ffi.PH_PRE_CV2)
L = _ffi.lib(); cyc = (ctypes.c_ulonglong * 8)()
for r in range(5): eng.forward_timed(B)
eng.sync(); L.ph_debug_conv_cycles(cyc, 1)
ms = eng.forward_timed(B); eng.sync(); L.ph_debug_conv_cycles(cyc, 0)
n = max(1, cyc[7])
print("op ms", [round(m, 4) for m in ms], "blocks", n)
print({k: round(cyc[i] / n) for i, k in enumerate(["frame-patch wait", "conv1_1", "K loop vmcnt wait (6 steps)", "K loop barrier wait (6 steps)", "start..end of K loop", "epilogue"])})
