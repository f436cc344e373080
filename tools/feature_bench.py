#!/usr/bin/env python3
"""Tracker appearance feature: the reference's way -- export all 78 upsampled heat maps of a frame
(datum.poseHeatMaps), crop under every torso box and resize_tensor to 25x50 on the host
(tracking/utils_track.py:45-57) -- against ph_pose_features, which samples only the needed values from the
resident 46x82 net output on the GPU.

    python tools/feature_bench.py [--batch 8] [--people 5] [--reps 5]
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pose-estimation_amd"))

from posehip import synth, utils_track  # noqa: E402
from posehip.engine import Engine  # noqa: E402
from posehip.tables import BODY_25, TORSO_IDS  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--people", type=int, default=5)
    ap.add_argument("--reps", type=int, default=5)
    args = ap.parse_args()
    eng = Engine(0)
    B, P = args.batch, args.people
    net = np.stack([synth.synth_net_output(300 + i, P, BODY_25, 46, 82) for i in range(B)])
    eng.set_net_out(net)
    res, _ = eng.decode_v2(BODY_25, B, 368, 656, 1.0, 128)
    counts = [len(r[1]) for r in res]
    win = [(0, 25), (25, 1), (26, 52)]
    t_dev, t_host, t_export = [], [], []
    for r in range(args.reps + 1):
        t0 = time.perf_counter()
        feat, status = eng.pose_features(B, 128, 25, TORSO_IDS, win, BODY_25.n_heat, 1, 368, 656, (656, 368), sum(counts))
        t1 = time.perf_counter()
        hms = [np.moveaxis(np.concatenate([eng.upsample_maps(b, c0, nc, 368, 656, BODY_25.n_heat, 1) for c0, nc in win]),
                           0, -1) for b in range(B)]
        t2 = time.perf_counter()
        bbox, _ = eng.pose_post(B, 128, 25, TORSO_IDS)
        dets = []       # per person, as the reference's loop; a degenerate crop raises IndexError there
        for b in range(B):
            for m in range(counts[b]):
                try:
                    dets += utils_track.create_detections(res[b][0][m:m + 1], res[b][1][m:m + 1], bbox[b, m:m + 1],
                                                          hms[b], [1.0, 1.0])
                except IndexError:
                    dets.append(None)
        t3 = time.perf_counter()
        if r:
            t_dev.append(t1 - t0)
            t_export.append(t2 - t1)
            t_host.append(t3 - t2)
    assert [d is None for d in dets] == (status < 0).tolist()
    for d, f in zip(dets, feat):
        assert d is None or np.array_equal(d.feature, f.reshape(-1).astype(np.float64))
    n = sum(counts)
    print(f"B={B} frames, {n} people: device features {1e3 * np.mean(t_dev):.2f} ms "
          f"({feat.nbytes / 1e6:.1f} MB to host) | reference way: export 78 maps/frame {1e3 * np.mean(t_export):.1f} ms "
          f"({sum(h.nbytes for h in hms) / 1e6:.0f} MB to host) + host crop/resize {1e3 * np.mean(t_host):.1f} ms; "
          f"identical features")
    eng.close()


if __name__ == "__main__":
    main()
