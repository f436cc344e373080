#!/usr/bin/env python3
"""Prints a hash of the network output of a few graphs with 96- and 128-channel 3x3 layers (and of the full BODY_25
graph), for comparing builds / switches across processes (the conv switches are read once per process):

    for e in X=1 POSEHIP_NO_LIN=1; do env $e python tools/split_check.py; done

The lines must be identical: the branch-free K loop and the generic loop accumulate in the same order.  (Written for the
weight / patch fetcher split of the 96-channel instantiation, DESIGN 9: equal hashes, no gain, removed.)"""
import hashlib, os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "pose-estimation_amd"))
from posehip import _ffi, caffe_io, graph, synth
from posehip.engine import Engine

h = hashlib.sha256()
eng = Engine(0)
for (H, W, B, chans) in ((46, 82, 32, (96, 96, 96)), (46, 82, 5, (288, 96, 96)), (23, 31, 7, (96, 96)), (40, 120, 3, (96, 96, 96)),
                         (46, 82, 32, (128, 128)), (33, 77, 2, (96, 128, 96))):
    txt = f'input: "image"\ninput_dim: 1\ninput_dim: 3\ninput_dim: {H}\ninput_dim: {W}\n'
    txt += caffe_io._conv("c0", "image", chans[0], 3) + caffe_io._act("ReLU", "r0", "c0")
    prev = "c0"
    for i, c in enumerate(chans):
        txt += caffe_io._conv(f"m{i}", prev, c, 3) + caffe_io._act("PReLU", f"p{i}", f"m{i}")
        prev = f"m{i}"
    txt += caffe_io._conv("out", prev, 24, 1)
    net = caffe_io.parse_prototxt(txt)
    plan = graph.build_plan(net, graph.random_weights(net, 7, out_gain=1.0))
    eng.build(plan, B)
    eng.preprocess(synth.synth_frames(3, B, H, W), _ffi.PH_PRE_CV2)
    for rep in range(3):
        eng.forward(B)
        out = eng.net_out(B)
        h.update(np.ascontiguousarray(out).tobytes())
net = caffe_io.parse_prototxt(caffe_io.body25_prototxt(368, 656))
plan = graph.build_plan(net, graph.random_weights(net, 0))
eng.build(plan, 8)
eng.preprocess(synth.synth_frames(1, 8, 368, 656), _ffi.PH_PRE_NATIVE)
for rep in range(3):
    eng.forward(8)
    h.update(np.ascontiguousarray(eng.net_out(8)).tobytes())
print(h.hexdigest(), sorted(set(eng.op_kernels())))
