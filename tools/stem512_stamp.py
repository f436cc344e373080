#!/usr/bin/env python3
"""Diagnostic build only: cycles per phase of the persistent stem kernel (conv_stem.hip) on the 368x656 batch-32 shape.

    make -C pose-estimation_amd BUILD=build_stamp LIB=lib/libposehip_stamp.so EXTRA=-DST_STAMP
    POSEHIP_LIB=pose-estimation_amd/lib/libposehip_stamp.so python tools/stem512_stamp.py
"""
import ctypes, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "pose-estimation_amd"))
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
L = _ffi.lib(); cyc = (ctypes.c_ulonglong * 16)()
for r in range(20): eng.forward_timed(B)
eng.sync(); L.ph_debug_stem512_cycles(cyc, 1)
ms = eng.forward_timed(B); eng.sync(); L.ph_debug_stem512_cycles(cyc, 0)
n = max(1, cyc[5])
print("op ms", [round(m, 4) for m in ms], "tiles", n)
names = ["P1 conv1_1 (frame pixels from LDS; rows stored, next loads issued)", "barrier A", "P2 K loop (+ next frame pixels -> LDS)", "barrier B", "P3 pool"]
tot = sum(cyc[i] for i in range(5))
for i, k in enumerate(names):
    print(f"{k:68s} {cyc[i] / n:9.0f} cycles per tile  {100 * cyc[i] / tot:5.1f} %")
print(f"{'sum':68s} {tot / n:9.0f}")
for i, k in ((6, "P1: row block 0"), (7, "P1: row blocks 1..4"), (8, "P1: next tile number (LDS), the one after (counter)"), (9, "P1: previous rows stored"), (10, "P1: next tile's 2 loads issued")):
    print(f"  {k:66s} {cyc[i] / n:9.0f}")
