#!/usr/bin/env python3
"""Randomised GPU-vs-oracle sweep (not collected by pytest; run by hand on a GPU box):

    python tests/fuzz_parity.py [--minutes 8] [--seed0 0]

V2 decode (both connectors) on synthetic people at random sizes / scales / crowding and on smoothed-noise maps
(hundreds of peaks, thousands of accepted pairs: exact ties, merges, conflicts); V1 decode on people and on
noise-topology heat maps (rings, nested islands, ragged blobs); both preprocess flavours at random source and net
sizes.  Everything is compared bit for bit, as in the
-m gpu tests; the first mismatch prints its recipe and the script exits 1.
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "pose-estimation_amd")):
    sys.path.insert(0, p)

from oracle import decode_v1, decode_v2  # noqa: E402
from posehip import synth  # noqa: E402
from posehip.engine import Engine  # noqa: E402
from posehip.tables import BODY_25, COCO_18, MAPIDX, get_body_parts_and_pose_pairs  # noqa: E402


def smooth_noise(rng, c, hs, ws, passes):
    f = rng.standard_normal((c, hs, ws)).astype(np.float32)
    for _ in range(passes):
        f = (np.roll(f, 1, 1) + 2 * f + np.roll(f, -1, 1)) / 4
        f = (np.roll(f, 1, 2) + 2 * f + np.roll(f, -1, 2)) / 4
    return f / f.std()


def v2_case(eng, seed):
    rng = np.random.default_rng(seed)
    base = BODY_25 if rng.random() < 0.7 else COCO_18
    conn = int(rng.integers(0, 2))
    model = base.with_connector(conn)
    hs, ws = [(46, 82), (46, 46), (23, 31), (30, 40), (17, 53)][int(rng.integers(0, 5))]
    kind = rng.random()
    if kind < 0.6:
        people = int(rng.integers(0, 31))
        frame = synth.synth_net_output(seed, people, model, hs, ws)
        what = f"people={people}"
    else:   # smoothed noise: peaks everywhere, PAFs that accept a large share of the pairs
        C = model.n_heat + 2 * model.n_pairs
        frame = np.zeros((C, hs, ws), np.float32)
        frame[:model.n_heat] = smooth_noise(rng, model.n_heat, hs, ws, 2) * 0.15 + 0.1
        frame[model.n_heat:] = smooth_noise(rng, 2 * model.n_pairs, hs, ws, 3) * float(rng.uniform(0.1, 0.6))
        what = "noise"
    up = int(rng.integers(4, 9))
    Ht, Wt = hs * up, ws * up
    scale = float(rng.choice([1.0, 1.75, 2.9433, 0.5]))
    eng.set_net_out(frame[None])
    recipe = f"V2 seed={seed} {base.name} connector={conn} {hs}x{ws} x{up} scale={scale} {what}"
    got, ex = eng.decode_v2(model, 1, Ht, Wt, scale=scale, want_peaks=True, want_subsets=True, max_people=1024)
    kp_o, sc_o, inter = decode_v2.decode(frame, model, Ht, Wt, scale, return_intermediates=True)
    assert np.array_equal(ex["peaks"][0], inter["peaks"]), recipe + ": peaks"
    rows = inter["rows"][inter["valid"]] if len(inter["valid"]) else np.zeros((0, model.n_parts + 1), np.int32)
    assert int(ex["n_people"][0]) == len(rows), recipe + f": {int(ex['n_people'][0])} people vs {len(rows)}"
    assert np.array_equal(ex["subsets"][0, :len(rows)], rows), recipe + ": subsets"
    assert np.array_equal(got[0][0], kp_o) and np.array_equal(got[0][1], sc_o), recipe + ": keypoints / scores"
    return recipe + f" -> {len(rows)} people"


BP, PP = get_body_parts_and_pose_pairs("COCO")


def v1_case(eng, seed):
    rng = np.random.default_rng(seed)
    hs, ws = [(46, 82), (46, 46), (23, 41), (16, 28)][int(rng.integers(0, 4))]
    W, H = [(656, 368), (480, 480), (333, 247), (960, 540), (1280, 720)][int(rng.integers(0, 5))]
    if rng.random() < 0.5:
        people = int(rng.integers(0, 41))
        pred = synth.synth_net_output(seed, people, COCO_18, hs, ws)[None]
        what = f"people={people}"
    else:
        parts = int(rng.integers(1, 7))
        pred = np.zeros((1, 57, hs, ws), np.float32)
        pred[0, :parts] = smooth_noise(rng, parts, hs, ws, int(rng.integers(1, 4))) * 0.2 + float(rng.uniform(-0.1, 0.3))
        pred[0, 19:] = smooth_noise(rng, 38, hs, ws, 3) * 0.3
        what = f"noise parts={parts}"
    recipe = f"V1 seed={seed} net {hs}x{ws} image {W}x{H} {what}"
    eng.set_net_out(pred)
    try:
        got = eng.decode_v1(BP, PP, MAPIDX, 1, W, H, 0.1, kp_cap=256, person_cap=1024)
    except Exception as e:   # capacity (-2) is a legal answer on the densest noise maps
        if "capacity" in str(e):
            return recipe + " -> capacity exceeded (skipped)"
        raise
    klist, people_g, ex = got[0]
    rk, rp, ri = decode_v1.postprocess_multi(pred, BP, PP, MAPIDX, W, H, 0.1, return_intermediates=True)
    assert klist.shape == rk.shape and np.array_equal(klist, rk), recipe + ": keypoints"
    assert ex["invalid_pairs"] == ri["invalid_pairs"], recipe + ": invalid pairs"
    for k, (a, b) in enumerate(zip(ex["valid_pairs"], ri["valid_pairs"])):
        if k not in ri["invalid_pairs"]:
            assert np.array_equal(np.asarray(a).reshape(-1, 3), np.asarray(b).reshape(-1, 3)), recipe + f": pairs {k}"
    assert len(people_g) == len(rp) and all(np.array_equal(a, b) for a, b in zip(people_g, rp)), recipe + ": people"
    return recipe + f" -> {rk.shape[0]} keypoints, {len(rp)} people"


def pre_case(eng, seed):
    """Both preprocess flavours at random source / net sizes (down-scales, up-scales -> cubic, ragged margins)."""
    from oracle import preprocess as pre_ref
    from posehip import _ffi, caffe_io, graph
    rng = np.random.default_rng(seed)
    Hn, Wn = 16 * int(rng.integers(4, 24)), 16 * int(rng.integers(4, 42))
    if rng.random() < 0.3:     # an up-scale
        H0, W0 = int(rng.integers(20, Hn)), int(rng.integers(20, Wn))
    else:
        H0, W0 = int(rng.integers(Hn // 2, 3 * Hn + 2)), int(rng.integers(Wn // 2, 3 * Wn + 2))
    frames = synth.synth_frames(seed, 1, H0, W0)
    net = caffe_io.parse_prototxt(f'input: "image" input_dim: 1 input_dim: 3 input_dim: {Hn} input_dim: {Wn}\n' +
                                  caffe_io._conv("out", "image", 8, 3))
    eng.build(graph.build_plan(net, graph.random_weights(net, 0)), 1)
    recipe = f"PRE seed={seed} source {H0}x{W0} -> net {Hn}x{Wn}"
    eng.preprocess(frames, _ffi.PH_PRE_NATIVE)
    got = eng.get_buffer(0)[:, 1:-1, 1:-1, :3]
    want = pre_ref.cv_mat_to_op_input(frames, Wn, Hn).transpose(0, 2, 3, 1)
    assert np.array_equal(got, want), recipe + f": native flavour, {int((got != want).sum())} values differ"
    eng.preprocess(frames, _ffi.PH_PRE_CV2)
    got = eng.get_buffer(0)[:, 1:-1, 1:-1, :3]
    ref = pre_ref.cv_resize_u8(frames, Wn, Hn).astype(np.float32) * np.float32(1 / 255.0)
    ref16 = ((ref.view(np.uint32) + 0x7fff + ((ref.view(np.uint32) >> 16) & 1)) >> 16 << 16).astype(np.uint32)
    assert np.array_equal(got.view(np.uint32), ref16), recipe + ": cv2 flavour"
    return recipe + " ok"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--minutes", type=float, default=8.0)
    ap.add_argument("--seed0", type=int, default=0)
    args = ap.parse_args()
    eng = Engine(0)
    t_end = time.time() + args.minutes * 60
    n = {"V1": 0, "V2": 0, "PRE": 0}
    seed = args.seed0
    while time.time() < t_end:
        for fn, key in ((v2_case, "V2"), (v1_case, "V1"), (pre_case, "PRE")):
            try:
                msg = fn(eng, seed)
            except AssertionError as e:
                print("MISMATCH", e, flush=True)
                sys.exit(1)
            n[key] += 1
            print(msg, flush=True)
        seed += 1
    print(f"fuzz ok: {n['V2']} V2 cases, {n['V1']} V1 cases, {n['PRE']} preprocess cases, seeds {args.seed0}..{seed - 1}", flush=True)


if __name__ == "__main__":
    main()
