#!/usr/bin/env python3
"""Single-layer timing of the MFMA implicit-GEMM conv kernel on the BODY_25 layer shapes.

    python tools/conv_bench.py [--batch 32] [--reps 20] [--shapes name,...]
Each shape is run as a 2-op net (1x1 conv 3->Cin feeding the measured conv) and the
second op is timed with HIP events (ph_forward_timed).
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pose-estimation_amd"))

from posehip import _ffi, caffe_io, graph, synth  # noqa: E402
from posehip.engine import Engine  # noqa: E402

SHAPES = {  # name: (H, W, Cin, Cout, k)
    "stem": (368, 656, 64, 64, 3),      # conv1_1 + conv1_2 + pool as the network runs them (conv_stem512_kernel)
    "conv1_2": (368, 656, 64, 64, 3),   # the same conv behind a 1x1 producer: plain 3x3 form, no stem fusion
    "conv2_2": (184, 328, 128, 128, 3),
    "conv3_2": (92, 164, 256, 256, 3),
    "conv4_2": (46, 82, 512, 512, 3),
    "m96": (46, 82, 96, 96, 3),
    "m288_96": (46, 82, 288, 96, 3),
    "m128": (46, 82, 128, 128, 3),
    "m384_128": (46, 82, 384, 128, 3),
    "m6_512": (46, 82, 384, 512, 1),
    "m7_52": (46, 82, 512, 52, 1),
    "pair512": (46, 82, 384, 512, 1, 52),   # Mconv6 + Mconv7 of a stage as the network runs them (conv1x1_pair_kernel;
    "pair256": (46, 82, 288, 256, 1, 52),   # POSEHIP_PAIR_KERNEL=0: the two launches)
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--shapes", type=str, default=",".join(SHAPES))
    ap.add_argument("--tile-n", type=int, default=0)
    args = ap.parse_args()
    eng = Engine(0)
    B = args.batch
    for name in args.shapes.split(","):
        H, W, cin, cout, k = SHAPES[name][:5]
        narrow = SHAPES[name][5] if len(SHAPES[name]) > 5 else 0
        if not narrow:   # a single layer is what is measured: its 8-channel consumer must not be fused into it
            os.environ["POSEHIP_PAIR_KERNEL"] = "0"
        else:
            os.environ.pop("POSEHIP_PAIR_KERNEL", None)
        txt = f'input: "image"\ninput_dim: 1\ninput_dim: 3\ninput_dim: {H}\ninput_dim: {W}\n'
        if name == "stem":
            txt += caffe_io._conv("c0", "image", cin, 3) + caffe_io._act("ReLU", "r0", "c0")
            txt += caffe_io._conv("c1", "c0", cout, k) + caffe_io._act("ReLU", "r1", "c1")
            txt += ('layer { name: "p1" type: "Pooling" bottom: "c1" top: "p1" pooling_param { pool: MAX kernel_size: 2 '
                    'stride: 2 } }\n')
            txt += caffe_io._conv("out", "p1", 8, 1)
        elif narrow:
            txt += caffe_io._conv("c0", "image", cin, 1) + caffe_io._act("ReLU", "r0", "c0")
            txt += caffe_io._conv("c1", "c0", cout, k) + caffe_io._act("PReLU", "r1", "c1")
            txt += caffe_io._conv("out", "c1", narrow, 1)
        else:
            txt += caffe_io._conv("c0", "image", cin, 1) + caffe_io._act("ReLU", "r0", "c0")
            txt += caffe_io._conv("c1", "c0", cout, k) + caffe_io._act("ReLU", "r1", "c1")
            txt += caffe_io._conv("out", "c1", 8, 1)
        net = caffe_io.parse_prototxt(txt)
        w = graph.random_weights(net, 0, out_gain=1.0)
        plan = graph.build_plan(net, w, tile_overrides={"c1": args.tile_n} if args.tile_n else None)
        eng.build(plan, B)
        eng.preprocess(synth.synth_frames(0, B, H, W), _ffi.PH_PRE_CV2)
        ts = []
        dbg = int(os.environ.get("POSEHIP_CONV_DEBUG", "0")) & 8
        import ctypes
        L = _ffi.lib()
        cyc = (ctypes.c_ulonglong * 8)()
        for r in range(args.reps + 3):
            if r == 3 and dbg:
                eng.sync()
                L.ph_debug_conv_cycles(cyc, 1)
            ms = eng.forward_timed(B)
            if r >= 3:
                ts.append(ms[1] + ms[2] if narrow else ms[1])
        if dbg:
            eng.sync()
            L.ph_debug_conv_cycles(cyc, 0)
            nb = max(1, cyc[4])
            print(f"  blocks {nb // args.reps}: prologue {cyc[0] / nb:.0f}  kloop {cyc[1] / nb:.0f} (wait {cyc[3] / nb:.0f})  epilogue {cyc[2] / nb:.0f} cycles/block")
        t = float(np.median(ts))
        fl = 2.0 * (k * k * cin * cout + cout * narrow) * H * W * B
        print(f"{name:10s} {H}x{W} {cin:4d}->{cout:4d}{f'->{narrow}' if narrow else ''} k{k}: {t:8.4f} ms  {fl / t / 1e9:8.1f} TFLOP/s",
              flush=True)


if __name__ == "__main__":
    main()
