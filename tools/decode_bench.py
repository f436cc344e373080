#!/usr/bin/env python3
"""Decode-only timing on synthetic people scenes (SURVEY 8d "decode stress"): a
random-weight network emits noise, so the decode kernels are also timed on
G2-style net outputs with P people per frame injected after the forward pass.

    python tools/decode_bench.py [--batch 32] [--reps 10]
Prints per-kernel device ms for V2 (BODY_25, 368x656) and V1 (COCO at 656x368 and 1920x1080).
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pose-estimation_amd"))

from posehip import synth  # noqa: E402
from posehip.engine import Engine  # noqa: E402
from posehip.tables import BODY_25, COCO_18, MAPIDX, get_body_parts_and_pose_pairs  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--people", type=str, default="1,5,20,60")
    ap.add_argument("--skip-v1", action="store_true")
    args = ap.parse_args()
    eng = Engine(0)
    B = args.batch
    bp, pp = get_body_parts_and_pose_pairs("COCO")
    for P in [int(v) for v in args.people.split(",")]:
        frames = np.stack([synth.synth_net_output(1000 * P + i, P, BODY_25, 46, 82) for i in range(B)])
        eng.set_net_out(frames)
        t = np.zeros(3)
        found = 0
        for r in range(args.reps + 2):
            res, ex = eng.decode_v2(BODY_25, B, 368, 656, max_people=128)
            if r >= 2:
                t += np.array(eng.decode_v2_times())
            found = int(ex["n_people"].sum())
        t /= args.reps
        print(f"V2 BODY_25 368x656  B={B} P={P:3d}: nms {t[0]:.3f} ms  paf {t[1]:.3f} ms  assemble {t[2]:.3f} ms  "
              f"total {t.sum():.3f} ms  ({found} skeletons)", flush=True)
    if args.skip_v1:
        return
    for (W, H, Bv) in ((656, 368, B), (1920, 1080, min(B, 8))):
        for P in [int(v) for v in args.people.split(",")]:
            frames = np.stack([synth.synth_net_output(2000 * P + i, P, COCO_18, 46, 82) for i in range(Bv)])
            eng.set_net_out(frames)
            t = np.zeros(3)
            for r in range(args.reps + 2):
                eng.decode_v1(bp, pp, MAPIDX, Bv, W, H, 0.1, kp_cap=256, person_cap=1024)
                if r >= 2:
                    t += np.array(eng.decode_v1_times())
            t /= args.reps
            print(f"V1 COCO {W}x{H}  B={Bv} P={P:3d}: runs {t[0]:.3f} ms  blobs {t[1]:.3f} ms  pairs+asm {t[2]:.3f} ms  "
                  f"total {t.sum():.3f} ms", flush=True)


if __name__ == "__main__":
    main()
