#!/usr/bin/env python3
"""BASELINE configs[2] end to end: OpenPose COCO-18 through the cv2.dnn flavour on crowded 1080p frames
(openpose/opencv/inference.py:57-89: predict + postprocess_multi per frame), batched.

    python tools/coco_mot17_bench.py [--batch 32] [--steps 10] [--people 20,60] [--json out.json]

One step = 32 resident 1920x1080 u8 frames -> blobFromImage resize to 656x368 -> COCO-18 forward (VGG + 6 stages of
7x7 convs) -> V1 decode at 1920x1080 (getKeypoints / getValidPairs / getPersonwiseKeypoints) -> host lists.
A random-weight network emits noise maps (thousands of blobs at the default threshold), so the decode runs on
synthetic P-people net outputs injected after the forward pass (SURVEY 8d "decode stress"); the forward is still
run and timed in every step.  Prints one JSON line per P: frames/s and the per-kernel milliseconds.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pose-estimation_amd"))

from posehip import _ffi, caffe_io, graph, synth  # noqa: E402
from posehip.engine import Engine  # noqa: E402
from posehip.tables import COCO_18, MAPIDX, get_body_parts_and_pose_pairs  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--people", type=str, default="20,60")
    ap.add_argument("--json", type=str, default="")
    ap.add_argument("--engines", type=int, default=3, help="engines (streams, host threads) of the in-flight leg; 0 = skip it")
    args = ap.parse_args()
    import torch
    H, W, H0, W0, B = 368, 656, 1080, 1920, args.batch
    net = caffe_io.parse_prototxt(caffe_io.coco_prototxt(H, W))
    plan = graph.build_plan(net, graph.random_weights(net, 0))
    eng = Engine(0)
    eng.build(plan, B)
    frames = torch.from_numpy(synth.synth_frames(5, B, H0, W0)).cuda()
    bp, pp = get_body_parts_and_pose_pairs("COCO")
    extra = []
    for _ in range(max(0, args.engines - 1)):   # the in-flight leg: one engine per host thread (ctypes calls drop the GIL)
        e2 = Engine(0)
        e2.build(plan, B)
        extra.append(e2)
    out = []
    for P in [int(v) for v in args.people.split(",")]:
        scene = torch.from_numpy(np.stack([synth.synth_net_output(3000 * P + i, P, COCO_18, 46, 82)
                                           for i in range(B)])).cuda()

        def step(timed=False):
            eng.preprocess(frames, _ffi.PH_PRE_CV2)
            if timed:
                eng.forward_timed(B, wait=False)
            else:
                eng.forward(B)
            eng.set_net_out(scene)
            return eng.decode_v1(bp, pp, MAPIDX, B, W0, H0, 0.1, kp_cap=256, person_cap=1024)

        for _ in range(2):
            res = step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            res = step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / args.steps
        fps_inflight = None
        if extra:
            import threading

            def worker(e, n):
                for _ in range(n):
                    e.preprocess(frames, _ffi.PH_PRE_CV2)
                    e.forward(B)
                    e.set_net_out(scene)
                    e.decode_v1(bp, pp, MAPIDX, B, W0, H0, 0.1, kp_cap=256, person_cap=1024)

            engs = [eng] + extra
            for e in extra:
                worker(e, 1)
            torch.cuda.synchronize()
            ths = [threading.Thread(target=worker, args=(e, args.steps)) for e in engs]
            t1 = time.perf_counter()
            for t in ths:
                t.start()
            for t in ths:
                t.join()
            torch.cuda.synchronize()
            fps_inflight = B * args.steps * len(engs) / (time.perf_counter() - t1)
        step(timed=True)
        fwd = eng.forward_times()
        v1 = eng.decode_v1_times()
        by = {}
        for n, t in zip(eng.op_kernels(), fwd):
            by[n] = by.get(n, 0.0) + float(t)
        line = {"config": "BASELINE configs[2]: COCO-18, 1920x1080 frames, cv2.dnn flavour, V1 decode at 1080p",
                "batch": B, "people_per_frame": P, "frames_per_s": round(B / dt, 1), "ms_per_step": round(dt * 1e3, 3),
                "frames_per_s_engines_in_flight": None if fps_inflight is None else round(fps_inflight, 1),
                "engines_in_flight": len(extra) + 1 if extra else 1,
                "forward_ms": round(float(fwd.sum()), 3),
                "forward_tflops": round(plan.flops_per_frame() * B / float(fwd.sum()) / 1e9, 1),
                "gflop_per_frame": round(plan.flops_per_frame() / 1e9, 1),
                "forward_kernels_ms": {k: round(v, 3) for k, v in sorted(by.items(), key=lambda kv: -kv[1])},
                "v1_runs_ms": round(v1[0], 3), "v1_blobs_ms": round(v1[1], 3), "v1_pairs_assemble_ms": round(v1[2], 3),
                "keypoints_per_frame": round(float(np.mean([len(r[0]) for r in res])), 1),
                "person_rows_per_frame": round(float(np.mean([len(r[1]) for r in res])), 1),
                "note": "frames_per_s / ms_per_step: wall clock of ONE engine, host list building of the V1 results included; "
                        "frames_per_s_engines_in_flight: the same steps from that many engines, one host thread each"}
        print(json.dumps(line), flush=True)
        out.append(line)
    if args.json:
        with open(args.json, "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
