#!/usr/bin/env python3
"""Do launches of the conv kernel from different HIP streams overlap?  A chain of L dependent 128->128 3x3 convs at
46x82 (the short layers of the BODY_25 stages), batch 32, on S engines (own stream each): wall time of R forward
passes in total, dealt round-robin to the engines, for S = 1, 2, 3.

    python tools/overlap_probe.py [--layers 20] [--reps 30]
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pose-estimation_amd"))

from posehip import _ffi, caffe_io, graph, synth  # noqa: E402
from posehip.engine import Engine  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--layers", type=int, default=20)
    ap.add_argument("--reps", type=int, default=30)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--cin", type=int, default=128)
    args = ap.parse_args()
    H, W, B, C = 46, 82, args.batch, args.cin
    txt = f'input: "image"\ninput_dim: 1\ninput_dim: 3\ninput_dim: {H}\ninput_dim: {W}\n'
    txt += caffe_io._conv("c0", "image", C, 1) + caffe_io._act("ReLU", "r0", "c0")
    prev = "c0"
    for i in range(args.layers):
        txt += caffe_io._conv(f"m{i}", prev, C, 3) + caffe_io._act("ReLU", f"rm{i}", f"m{i}")
        prev = f"m{i}"
    txt += caffe_io._conv("out", prev, 8, 1)
    net = caffe_io.parse_prototxt(txt)
    plan = graph.build_plan(net, graph.random_weights(net, 0, out_gain=1.0))
    frames = synth.synth_frames(0, B, H, W)
    engs = []
    for s in range(3):
        e = Engine(0)
        e.build(plan, B)
        e.preprocess(frames, _ffi.PH_PRE_CV2)
        e.forward(B)
        e.sync()
        engs.append(e)
    gflop = 2 * 9 * C * C * H * W * B * args.layers / 1e9
    for S in (1, 2, 3, 1):
        for e in engs:
            e.sync()
        t0 = time.perf_counter()
        for r in range(args.reps):
            engs[r % S].forward(B)
        for e in engs:
            e.sync()
        dt = time.perf_counter() - t0
        print(f"S={S}: {dt / args.reps * 1e3:.3f} ms per pass of {args.layers} layers = "
              f"{dt / args.reps / args.layers * 1e6:.1f} us per layer, {gflop * args.reps / dt / 1e3:.0f} TFLOP/s", flush=True)


if __name__ == "__main__":
    main()
