#!/usr/bin/env python3
"""Randomised GPU-vs-oracle sweep (not collected by pytest; run by hand on a GPU box):

    python tests/fuzz_parity.py [--minutes 8] [--seed0 0]

V2 decode (both connectors) on synthetic people at random sizes / scales / crowding and on smoothed-noise maps
(hundreds of peaks, thousands of accepted pairs: exact ties, merges, conflicts); V1 decode on people and on
noise-topology heat maps (rings, nested islands, ragged blobs); both preprocess flavours at random source and net
sizes; random small Caffe graphs through the plan builder and the conv kernels (< 1e-2 of the fp32 oracle with bf16
storage emulated).  Everything else is compared bit for bit, as in the
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
    # (the last two: two / four output columns per lane in the NMS kernel)
    hs, ws = [(46, 82), (46, 46), (23, 31), (30, 40), (17, 53), (12, 130), (14, 260)][int(rng.integers(0, 7))]
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
    if rng.random() < 0.3:     # ratios that are not integers: runs of rows with equal vertical taps of uneven length
        Ht, Wt = Ht + int(rng.integers(0, 4)), Wt + int(rng.integers(0, 4))
    scale = float(rng.choice([1.0, 1.75, 2.9433, 0.5]))
    eng.set_net_out(frame[None])
    recipe = f"V2 seed={seed} {base.name} connector={conn} {hs}x{ws} -> {Ht}x{Wt} scale={scale} {what}"
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
    group = seed % 4 == 3     # every fourth case through the one-group walk of the irregular blobs only
    recipe = f"V1 seed={seed} net {hs}x{ws} image {W}x{H} {what}" + (" (one-group walk)" if group else "")
    eng.set_net_out(pred)
    try:
        if group:
            os.environ["POSEHIP_V1_FAST"] = "0"
        got = eng.decode_v1(BP, PP, MAPIDX, 1, W, H, 0.1, kp_cap=256, person_cap=1024)
    except Exception as e:   # capacity (-2) is a legal answer on the densest noise maps
        if "capacity" in str(e):
            return recipe + " -> capacity exceeded (skipped)"
        raise
    finally:
        os.environ.pop("POSEHIP_V1_FAST", None)
    wave_n, group_n = eng.decode_v1_walks()
    assert not (group and wave_n), recipe
    recipe += f" [{wave_n} blobs by a wave, {group_n} by one group]"
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


def net_case(eng, seed):
    """A random small Caffe graph (3x3 / 1x1 / 7x7 convs, ReLU / PReLU, 2x2 pools, concats of same-size blobs,
    odd channel counts and map sizes) through graph.build_plan + the conv kernels vs the torch fp32 oracle."""
    from oracle import net_ref
    from posehip import _ffi, caffe_io, graph
    rng = np.random.default_rng(seed)
    h, w = int(rng.integers(9, 70)), int(rng.integers(9, 180))
    txt = f'input: "image"\ninput_dim: 1\ninput_dim: 3\ninput_dim: {h}\ninput_dim: {w}\n'
    widths = [8, 16, 19, 26, 32, 38, 40, 52, 64, 96, 128, 160]
    blobs = [("image", 3, 0)]          # (name, channels, pooling level)
    level, desc = 0, []
    n_layers = int(rng.integers(2, 8))
    for i in range(n_layers):
        same = [b for b in blobs if b[2] == level and b[0] != "image"]
        kind = rng.random()
        plain = [b for b in same if not b[0].startswith("cat")]   # (the planner rejects a concat of concats, loudly)
        if kind < 0.15 and len(plain) >= 2:
            k = int(rng.integers(2, min(4, len(plain)) + 1))
            members = [plain[j] for j in sorted(rng.choice(len(plain), k, replace=False))]
            name = f"cat{i}"
            txt += caffe_io._concat(name, [m[0] for m in members])
            blobs.append((name, sum(m[1] for m in members), level))
            desc.append(f"concat{[m[1] for m in members]}")
            continue
        bottom = blobs[-1] if blobs[-1][2] == level else same[-1]
        if kind < 0.3 and bottom[0] != "image" and (h >> level) >= 6 and (w >> level) >= 6:
            name = f"p{i}"
            txt += caffe_io._pool(name, bottom[0])
            level += 1
            blobs.append((name, bottom[1], level))
            desc.append("pool")
            continue
        cout = int(rng.choice(widths))
        k = int(rng.choice([3, 3, 3, 1, 1, 7])) if bottom[0] != "image" else 3
        name = f"c{i}"
        txt += caffe_io._conv(name, bottom[0], cout, k)
        act = [None, "ReLU", "PReLU"][int(rng.integers(0, 3))]
        if act:
            txt += caffe_io._act(act, "a" + name, name)
        blobs.append((name, cout, level))
        desc.append(f"conv{k}x{k}:{bottom[1]}->{cout}{'+' + act if act else ''}")
    last = blobs[-1] if blobs[-1][0] != "image" else None
    if last is None or last[0].startswith("p") or last[0].startswith("cat"):
        txt += caffe_io._conv("out", blobs[-1][0], int(rng.choice([8, 19, 26])), 1 if blobs[-1][0] != "image" else 3)
        desc.append("conv:out")
    batch = int(rng.integers(1, 4))
    recipe = f"NET seed={seed} {h}x{w} batch={batch} " + " ".join(desc)
    net = caffe_io.parse_prototxt(txt)
    weights = graph.random_weights(net, seed=seed, out_gain=1.0)
    try:
        plan = graph.build_plan(net, weights)
    except ValueError as e:      # a graph shape the planner refuses by design (pooled concat, ...): loud, not wrong
        return recipe + f" -> refused by the planner ({e})"
    eng.build(plan, batch)
    x = rng.integers(0, 256, (batch, h, w, 3), dtype=np.uint8)
    B = eng.preprocess(x, _ffi.PH_PRE_CV2)
    eng.forward(B)
    got = eng.net_out(B)
    xf = x.astype(np.float32).transpose(0, 3, 1, 2) * np.float32(1.0 / 255.0)
    ref16 = net_ref.forward(net, weights, xf, bf16_sim=True)
    assert got.shape == ref16.shape, recipe + f": shape {got.shape} vs {ref16.shape}"
    scale = float(np.abs(ref16).max()) + 1e-6
    err = float(np.abs(got - ref16).max()) / scale
    assert np.isfinite(got).all() and err < 1e-2, recipe + f": relative error {err:.3e}"
    return recipe + f" -> err {err:.1e}"


def post_case(eng, seed):
    """What the estimator derives from the keypoints on the device: torso boxes, depth lift (fp64), heat-map export
    under --heatmaps_scale, and the tracker feature (crop under the torso box + resize_tensor), all bit for bit."""
    from oracle import features, post_v2
    from posehip.tables import TORSO_IDS
    rng = np.random.default_rng(seed)
    hs, ws = [(46, 82), (46, 46), (23, 31)][int(rng.integers(0, 3))]
    Ht, Wt = hs * 8, ws * 8
    B = int(rng.integers(1, 4))
    people = [int(rng.integers(0, 9)) for _ in range(B)]
    net = np.stack([synth.synth_net_output(seed * 7 + i, people[i], BODY_25, hs, ws) for i in range(B)])
    eng.set_net_out(net)
    H0, W0 = int(rng.integers(Ht, 3 * Ht)), int(rng.integers(Wt, 3 * Wt))
    scale = float(rng.choice([1.0, H0 / Ht, 2.9433]))
    recipe = f"POST seed={seed} net {hs}x{ws} B={B} people={people} image {H0}x{W0} scale={scale:.4f}"
    res, ex = eng.decode_v2(BODY_25, B, Ht, Wt, scale, 128)
    depth = rng.integers(0, 6000, (B, H0, W0), dtype=np.uint16)
    depth[rng.random(depth.shape) < 0.05] = 0                      # holes in the depth image
    intr = (float(rng.uniform(500, 1200)), float(rng.uniform(500, 1200)), W0 / 2 + float(rng.uniform(-20, 20)),
            H0 / 2 + float(rng.uniform(-20, 20)))
    patch = int(rng.integers(0, 4))
    bbox, kp3d = eng.pose_post(B, 128, BODY_25.n_parts, TORSO_IDS, depth, intr, 1e-3, patch)
    counts = []
    for b in range(B):
        kp = res[b][0]
        m = len(kp)
        counts.append(m)
        assert np.array_equal(bbox[b, :m], post_v2.torso_bbox(kp, TORSO_IDS)), recipe + f": torso boxes frame {b}"
        for q in range(m):
            want = post_v2.lift_3d(kp[q], depth[b], *intr, 1e-3, patch)
            assert np.array_equal(kp3d[b, q].view(np.uint64), want.view(np.uint64)), recipe + f": 3-D lift {b}/{q}"
    mode = int(rng.integers(0, 4))
    windows = [[(0, 25)], [(0, 25), (25, 1), (26, 52)], [(26, 52)], [(3, 9), (25, 1)]][int(rng.integers(0, 4))]
    size = [(25, 50), (7, 4), (13, 31)][int(rng.integers(0, 3))]
    if sum(counts):
        feat, status = eng.pose_features(B, 128, BODY_25.n_parts, TORSO_IDS, windows, BODY_25.n_heat, mode, Ht, Wt,
                                         (W0, H0), sum(counts), size)
        row = 0
        for b in range(B):
            if not counts[b]:
                continue
            hm = np.moveaxis(features.export_heatmaps(net[b], windows, BODY_25.n_heat, mode, Ht, Wt), 0, -1)
            want = features.detection_features(post_v2.torso_bbox(res[b][0], TORSO_IDS), hm, (W0, H0), size)
            for w in want:
                if isinstance(w, Exception):
                    assert status[row] == -1 and not feat[row].any(), recipe + f": feature row {row} should be refused"
                else:
                    assert status[row] == 1 and np.array_equal(feat[row].reshape(-1).astype(np.float64), w), \
                        recipe + f": feature row {row} mode {mode} windows {windows} size {size}"
                row += 1
    return recipe + f" -> {sum(counts)} people, mode {mode}"


def movenet_case(eng, seed):
    """MoveNet steps around the model: crop_and_resize (int32), coordinate update + next crop region (float64), and
    the MultiPose head decode, at random sizes / regions / scores."""
    from oracle import movenet as omv
    from posehip import movenet as mv
    rng = np.random.default_rng(seed)
    B = int(rng.integers(1, 6))
    ih, iw = int(rng.integers(40, 300)), int(rng.integers(40, 400))
    frames = rng.integers(0, 256, (B, ih, iw, 3), dtype=np.uint8)
    regions = []
    for b in range(B):
        if rng.random() < 0.3:
            regions.append(omv.init_crop_region(ih, iw))
        else:
            y0, x0 = rng.uniform(-0.4, 0.7, 2)
            side = rng.uniform(0.1, 1.4)
            asp = rng.uniform(0.5, 1.5)
            regions.append({'y_min': y0, 'x_min': x0, 'y_max': y0 + side, 'x_max': x0 + side * asp, 'height': side,
                            'width': side * asp})
    size = int(rng.integers(1, 70))
    recipe = f"MOVENET seed={seed} B={B} image {ih}x{iw} crop {size}"
    got = mv.crop_and_resize_batch(frames, regions, (size, size))
    for b in range(B):
        assert np.array_equal(got[b], omv.crop_and_resize(frames[b], regions[b], (size, size))), recipe + f": crop {b}"
    k = rng.uniform(-0.1, 1.1, (B, 1, 17, 3)).astype(np.float32)
    k[:, :, :, 2] = rng.choice([0.0, 0.05, 0.19, 0.2, 0.21, 0.6, 0.95], (B, 1, 17)).astype(np.float32)
    got_k, got_r = mv.update_and_determine_batch(k, regions, ih, iw)
    for b in range(B):
        want_k = omv.update_keypoints(k[b:b + 1], regions[b], ih, iw)
        assert np.array_equal(got_k[b:b + 1].view(np.uint32), want_k.view(np.uint32)), recipe + f": keypoints {b}"
        want_r = omv.determine_crop_region(want_k, ih, iw)
        for key in want_r:
            assert np.float64(got_r[b][key]).view(np.uint64) == np.float64(want_r[key]).view(np.uint64), \
                recipe + f": crop region {b} {key}"
    G, P = int(rng.integers(4, 70)), int(rng.integers(1, 13))
    Bm = int(rng.integers(1, 4))
    center = rng.uniform(0, 1, (Bm, G, G)).astype(np.float32) ** int(rng.integers(1, 6))
    heat = rng.uniform(0, 1, (Bm, G, G, 17)).astype(np.float32)
    regr = rng.normal(0, G / 8, (Bm, G, G, 34)).astype(np.float32)
    offs = rng.uniform(-0.5, 0.5, (Bm, G, G, 34)).astype(np.float32)
    box = rng.uniform(1, G / 2 + 1, (Bm, G, G, 2)).astype(np.float32)
    thr = float(rng.choice([0.1, 0.3, 0.9]))
    gotd = mv.multipose_decode(center, heat, regr, offs, box, P, thr)
    for b in range(Bm):
        want = omv.multipose_decode(center[b], heat[b], regr[b], offs[b], box[b], P, thr)
        assert np.array_equal(gotd[b].view(np.uint32), want.view(np.uint32)), recipe + f": head decode {b} G={G} P={P} thr={thr}"
    return recipe + f" G={G} P={P} ok"


def stem_case(eng, seed):
    """Round 3: the persistent stem kernel (conv1_1 + conv1_2 + pool) against the halo kernel's stem form, bit for bit,
    at random even sizes and batches (ragged tiles in both directions, single-tile images, several images per block)."""
    from posehip import caffe_io, graph, _ffi
    rng = np.random.default_rng(seed)
    H, W, B = 2 * int(rng.integers(4, 120)), 2 * int(rng.integers(4, 180)), int(rng.integers(1, 4))
    txt = f'input: "image"\ninput_dim: 1\ninput_dim: 3\ninput_dim: {H}\ninput_dim: {W}\n'
    txt += caffe_io._conv("c11", "image", 64, 3) + caffe_io._act("ReLU", "r11", "c11")
    txt += caffe_io._conv("c12", "c11", 64, 3) + caffe_io._act("ReLU" if rng.random() < 0.8 else "PReLU", "r12", "c12")
    txt += 'layer { name: "p1" type: "Pooling" bottom: "c12" top: "p1" pooling_param { pool: MAX kernel_size: 2 stride: 2 } }\n'
    txt += caffe_io._conv("out", "p1", 8, 1)
    net = caffe_io.parse_prototxt(txt)
    plan = graph.build_plan(net, graph.random_weights(net, seed, out_gain=1.0))
    frames = synth.synth_frames(seed, B, H, W)
    got = []
    for flag in ("0", "1"):
        os.environ["POSEHIP_STEM_KERNEL"] = flag
        eng.build(plan, B)
        assert (eng.op_kernels()[1].startswith("conv_stem512")) == (flag == "1"), eng.op_kernels()
        for _ in range(2):
            eng.preprocess(frames, _ffi.PH_PRE_NATIVE)
            eng.forward(B)
        got.append(eng.get_buffer(plan.ops[1]["dst"][0][0]))
    os.environ.pop("POSEHIP_STEM_KERNEL", None)
    recipe = f"STEM seed={seed} {H}x{W} B={B}"
    assert np.array_equal(got[0], got[1]) and np.abs(got[1]).max() > 0, recipe
    return recipe + " ok"


def pair_case(eng, seed):
    """Round 3: conv1x1_pair_kernel (a wide 1x1 conv + the narrow 1x1 conv that alone reads it, one launch) against the
    two separate launches, bit for bit: random input widths (incl. odd chunk counts), 1-4 slabs, any narrow width up to
    64, every activation pair, ragged and sub-tile pixel counts; the narrow output both as fp32 network output and as
    the bf16 window a further 3x3 conv reads."""
    from posehip import caffe_io, graph, _ffi
    rng = np.random.default_rng(seed)
    H, W, B = int(rng.integers(3, 60)), int(rng.integers(3, 90)), int(rng.integers(1, 4))
    cin = 32 * int(rng.integers(3, 13))
    wide = 128 * int(rng.integers(1, 5))
    narrow = int(rng.integers(1, 65))
    acts = [None, "ReLU", "PReLU"]
    a1, a2 = acts[int(rng.integers(0, 3))], acts[int(rng.integers(0, 3))]
    txt = f'input: "image"\ninput_dim: 1\ninput_dim: 3\ninput_dim: {H}\ninput_dim: {W}\n'
    txt += caffe_io._conv("f", "image", cin, 3) + caffe_io._act("ReLU", "rf", "f")
    txt += caffe_io._conv("m6", "f", wide, 1) + (caffe_io._act(a1, "r6", "m6") if a1 else "")
    txt += caffe_io._conv("m7", "m6", narrow, 1) + (caffe_io._act(a2, "r7", "m7") if a2 else "")
    txt += caffe_io._concat("cat", ["f", "m7"])
    txt += caffe_io._conv("tail", "cat", 16, 3)
    txt += caffe_io._concat("net_output", ["m7", "tail"])
    net = caffe_io.parse_prototxt(txt)
    plan = graph.build_plan(net, graph.random_weights(net, seed, out_gain=1.0))
    eng.build(plan, B)
    frames = synth.synth_frames(seed, B, H, W)
    got = []
    for flag in ("0", "1"):
        os.environ["POSEHIP_PAIR_KERNEL"] = flag
        assert ("conv1x1_pair_kernel(PhPairParams)" in eng.op_kernels()) == (flag == "1"), eng.op_kernels()
        eng.preprocess(frames, _ffi.PH_PRE_NATIVE)
        eng.forward(B)
        got.append(eng.net_out(B))
    os.environ.pop("POSEHIP_PAIR_KERNEL", None)
    recipe = f"PAIR seed={seed} {H}x{W} B={B} {cin}->{wide}({a1})->{narrow}({a2})"
    assert np.array_equal(got[0], got[1]) and np.isfinite(got[1]).all() and np.abs(got[1]).max() > 0, recipe
    return recipe + " ok"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--minutes", type=float, default=8.0)
    ap.add_argument("--seed0", type=int, default=0)
    ap.add_argument("--only", type=str, default="", help="comma list of V2,V1,PRE,NET,POST,MOVENET,STEM,PAIR")
    args = ap.parse_args()
    eng = Engine(0)
    t_end = time.time() + args.minutes * 60
    n = {"V1": 0, "V2": 0, "PRE": 0, "NET": 0, "POST": 0, "MOVENET": 0, "STEM": 0, "PAIR": 0}
    seed = args.seed0
    while time.time() < t_end:
        cases = ((v2_case, "V2"), (v1_case, "V1"), (pre_case, "PRE"), (net_case, "NET"), (post_case, "POST"), (movenet_case, "MOVENET"), (stem_case, "STEM"), (pair_case, "PAIR"))
        if args.only:
            cases = tuple(c for c in cases if c[1] in args.only.split(","))
        for fn, key in cases:
            try:
                msg = fn(eng, seed)
            except AssertionError as e:
                print("MISMATCH", e, flush=True)
                sys.exit(1)
            n[key] += 1
            print(msg, flush=True)
        seed += 1
    print(f"fuzz ok: {n['V2']} V2 cases, {n['V1']} V1 cases, {n['PRE']} preprocess cases, {n['NET']} network cases, {n['POST']} post-processing cases, {n['MOVENET']} MoveNet cases, {n['STEM']} stem-kernel cases, {n['PAIR']} pair-kernel cases, seeds {args.seed0}..{seed - 1}", flush=True)


if __name__ == "__main__":
    main()
