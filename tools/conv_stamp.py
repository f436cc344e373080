#!/usr/bin/env python3
"""Diagnostic build only: per-phase cycle stamps of the halo kernel's branch-free K loop (conv4_2 shape) and the
in-kernel clock (s_memtime / s_memrealtime).  Needs the library built with -DPH_LIN_STAMP:

    make -C pose-estimation_amd clean all CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -DPH_LIN_STAMP"
    python tools/conv_stamp.py [cin cout]
(the shipped library has no stamps; rebuild it normally afterwards)."""
import ctypes, os, sys
sys.path.insert(0, "pose-estimation_amd")
import numpy as np
from posehip import _ffi, caffe_io, graph, synth
from posehip.engine import Engine
eng = Engine(0); B = 32
H, W, k = 46, 82, 3
cin = int(sys.argv[1]) if len(sys.argv) > 1 else 512
cout = int(sys.argv[2]) if len(sys.argv) > 2 else 512
txt = f'input: "image"\ninput_dim: 1\ninput_dim: 3\ninput_dim: {H}\ninput_dim: {W}\n'
txt += caffe_io._conv("c0", "image", cin, 1) + caffe_io._act("ReLU", "r0", "c0")
txt += caffe_io._conv("c1", "c0", cout, k) + caffe_io._act("ReLU", "r1", "c1")
txt += caffe_io._conv("out", "c1", 8, 1)
net = caffe_io.parse_prototxt(txt)
plan = graph.build_plan(net, graph.random_weights(net, 0, out_gain=1.0))
eng.build(plan, B)
eng.preprocess(synth.synth_frames(0, B, H, W), _ffi.PH_PRE_CV2)
L = _ffi.lib(); cyc = (ctypes.c_ulonglong * 8)()
import time
t0 = time.time()
while time.time() - t0 < 2.5: eng.forward_timed(B)
eng.sync(); L.ph_debug_conv_cycles(cyc, 1)
ms = eng.forward_timed(B); eng.sync(); L.ph_debug_conv_cycles(cyc, 0)
nblk = 504 * ((cout + 127) // 128)
n = nblk * (cin // 32) * 9
names = ["barrier wait", "addr+reads issue", "dma issue", "lgkm wait", "mfma issue", "vmcnt wait"]
print("kernel ms", ms[1], " per step (wave 1 of every block):", {names[i]: round(cyc[i] / n, 1) for i in range(6)}, " loop cycles/block", cyc[6] // nblk, " in-kernel clock GHz", round(cyc[6] / max(1, cyc[7]) * 0.1, 3))
if os.environ.get("POSEHIP_STAMP_EPI"):
    nb = max(1, cyc[4])
    print("epilogue per block:", {k: round(cyc[i] / nb) for i, k in enumerate(["bias+act", "barrier+row table", "staging+barrier", "store loop"])}, "blocks", nb)
