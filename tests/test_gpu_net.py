"""GPU parity: MFMA implicit-GEMM conv / pool / concat plan vs the torch fp32
oracle (oracle/net_ref.py), through the C ABI."""
import numpy as np
import pytest

from posehip import _ffi, caffe_io, graph

pytestmark = pytest.mark.gpu


def _tiny_net(h, w, spec):
    """spec: list of ('conv', name, bottom, cout, k, act) | ('pool', name, bottom) | ('concat', name, [bottoms])."""
    s = f'input: "image"\ninput_dim: 1\ninput_dim: 3\ninput_dim: {h}\ninput_dim: {w}\n'
    for it in spec:
        if it[0] == "conv":
            _, name, bottom, cout, k, act = it
            s += caffe_io._conv(name, bottom, cout, k)
            if act:
                s += caffe_io._act(act, "act_" + name, name)
        elif it[0] == "pool":
            s += caffe_io._pool(it[1], it[2])
        else:
            s += caffe_io._concat(it[1], it[2])
    return caffe_io.parse_prototxt(s)


def _run(engine, net, weights, x_u8, batch):
    from oracle import net_ref
    plan = graph.build_plan(net, weights)
    engine.build(plan, batch)
    B = engine.preprocess(x_u8, _ffi.PH_PRE_CV2)
    engine.forward(B)
    got = engine.net_out(B)
    x = x_u8.astype(np.float32).transpose(0, 3, 1, 2) * np.float32(1.0 / 255.0)
    ref16 = net_ref.forward(net, weights, x, bf16_sim=True)
    ref32 = net_ref.forward(net, weights, x, bf16_sim=False)
    return got, ref16, ref32


CASES = {
    # exercises: Cin=3 first layer, 3x3, ReLU, pool, PReLU, 1x1, concat-by-offset with
    # unaligned member widths (26/52 -> 8-aligned windows), 7x7, odd spatial sizes, M tails
    "vgg_like": (24, 40, [("conv", "c1", "image", 64, 3, "ReLU"), ("conv", "c2", "c1", 64, 3, "ReLU"),
                           ("pool", "p1", "c2"), ("conv", "c3", "p1", 128, 3, "PReLU"),
                           ("conv", "out", "c3", 26, 1, None)]),
    "dense_concat": (16, 24, [("conv", "f", "image", 128, 3, "PReLU"),
                               ("conv", "a0", "f", 96, 3, "PReLU"), ("conv", "a1", "a0", 96, 3, "PReLU"),
                               ("conv", "a2", "a1", 96, 3, "PReLU"), ("concat", "cat1", ["a0", "a1", "a2"]),
                               ("conv", "m6", "cat1", 256, 1, "PReLU"), ("conv", "l2", "m6", 52, 1, None),
                               ("concat", "cat2", ["f", "l2"]),
                               ("conv", "b0", "cat2", 128, 3, "PReLU"), ("conv", "l1", "b0", 26, 1, None),
                               ("concat", "cat3", ["f", "l1", "l2"]),
                               ("conv", "c0", "cat3", 96, 3, "ReLU"), ("conv", "l1b", "c0", 26, 1, None),
                               ("concat", "net_output", ["l1b", "l2"])]),
    "k7": (14, 18, [("conv", "f", "image", 64, 3, "ReLU"), ("conv", "g", "f", 128, 7, "ReLU"),
                     ("conv", "out", "g", 19, 1, None)]),
    # width > 126: the 3x3 halo kernel switches from padded-linear runs to 16x16 tiles (ragged right/bottom tiles)
    "wide_tiles": (18, 150, [("conv", "c1", "image", 32, 3, "ReLU"), ("conv", "c2", "c1", 64, 3, "PReLU"),
                              ("conv", "c3", "c2", 96, 3, "ReLU"), ("conv", "out", "c3", 20, 1, None)]),
    # single-chunk K loops: 1x1 over 32 channels (one K step), 3x3 over 32 channels (one patch, nine steps)
    "one_chunk": (20, 36, [("conv", "a", "image", 32, 3, "ReLU"), ("conv", "b", "a", 40, 3, "PReLU"),
                            ("conv", "c", "b", 32, 1, "ReLU"), ("conv", "out", "c", 20, 1, None)]),
    # 1x1 layers on a large map: the epilogue's row table divides by precomputed reciprocals at batch 1 and falls
    # back to real divisions at batch 3 (pixel index x image size beyond 2^32: conv_mfma.hip, ph_epilogue_magic)
    "big_1x1": (200, 300, [("conv", "c1", "image", 32, 3, "ReLU"), ("conv", "c2", "c1", 32, 1, "PReLU"),
                            ("conv", "out", "c2", 20, 1, None)]),
    # a pooled map that is a member of a concat: the pool op writes a channel window of the wider buffer
    # (found by tests/fuzz_parity.py; not a shape the OpenPose graphs use)
    "pool_into_concat": (17, 30, [("conv", "c0", "image", 38, 3, "PReLU"), ("pool", "p1", "c0"),
                                   ("conv", "c2", "p1", 8, 1, None), ("concat", "cat3", ["p1", "c2"]),
                                   ("conv", "c4", "cat3", 128, 3, None), ("conv", "out", "c4", 40, 7, None)]),
    "odd_pool": (13, 19, [("conv", "c1", "image", 64, 3, "ReLU"), ("pool", "p", "c1"),
                           ("conv", "out", "p", 38, 3, None)]),
}


@pytest.mark.parametrize("case", sorted(CASES))
@pytest.mark.parametrize("batch", [1, 3])
def test_small_nets_match_oracle(engine, case, batch):
    h, w, spec = CASES[case]
    net = _tiny_net(h, w, spec)
    weights = graph.random_weights(net, seed=7, out_gain=1.0)
    rng = np.random.default_rng(3)
    x = rng.integers(0, 256, (batch, h, w, 3), dtype=np.uint8)
    got, ref16, ref32 = _run(engine, net, weights, x, batch)
    assert got.shape == ref32.shape
    scale = np.abs(ref32).max() + 1e-6
    # vs bf16-storage emulation: only fp32 accumulation-order noise (+ rare 1-ulp bf16 flips)
    assert np.abs(got - ref16).max() / scale < 1e-2, np.abs(got - ref16).max() / scale
    # vs pure fp32: bf16 storage of weights/activations, a few layers deep
    assert np.abs(got - ref32).max() / scale < 3e-2, np.abs(got - ref32).max() / scale
    assert np.isfinite(got).all()


def test_forward_is_deterministic_and_batch_invariant(engine):
    h, w, spec = CASES["dense_concat"]
    net = _tiny_net(h, w, spec)
    weights = graph.random_weights(net, seed=1, out_gain=1.0)
    plan = graph.build_plan(net, weights)
    engine.build(plan, 4)
    rng = np.random.default_rng(0)
    x = rng.integers(0, 256, (4, h, w, 3), dtype=np.uint8)
    engine.preprocess(x, _ffi.PH_PRE_CV2)
    engine.forward(4)
    a = engine.net_out(4)
    engine.preprocess(x[2:3], _ffi.PH_PRE_CV2)
    engine.forward(1)
    b = engine.net_out(1)
    assert np.array_equal(a[2], b[0])          # frames are independent units
    engine.preprocess(x, _ffi.PH_PRE_CV2)
    engine.forward(4)
    assert np.array_equal(a, engine.net_out(4))


def test_bad_plans_fail_loudly(engine):
    net = _tiny_net(8, 8, [("conv", "out", "image", 8, 3, None)])
    weights = graph.random_weights(net, seed=0)
    plan = graph.build_plan(net, weights)
    plan.ops[0]["in_coff"] = 4                  # misaligned channel window
    with pytest.raises(_ffi.PosehipError):
        engine.build(plan, 1)
    with pytest.raises(_ffi.PosehipError):
        engine.forward(1)                        # nothing built after the failure


def test_body25_full_size_matches_oracle_and_is_batch_invariant(engine):
    """BASELINE configs[1] geometry (BODY_25, 368x656): every production kernel path at its real shapes
    (stem fusion, fused pooling, 16x16-tile and padded-linear halo modes, 1x1 pipeline, fp32 epilogue).
    One frame is checked against the torch fp32 oracle (north_star tolerance 1e-2 on the heat maps); frames
    are independent units, so frame 1 of a batch of 3 must equal the same frame run alone bit for bit; the
    V2 decode of the GPU's own network output must equal the oracle's decode of that output bit for bit."""
    from oracle import decode_v2, net_ref
    from posehip import synth
    from posehip.tables import BODY_25
    H, W = 368, 656
    net = caffe_io.parse_prototxt(caffe_io.body25_prototxt(H, W))
    weights = graph.random_weights(net, seed=5)
    plan = graph.build_plan(net, weights)
    assert abs(plan.flops_per_frame() / 1e9 - 287.26) < 0.05
    engine.build(plan, 3)
    frames = synth.synth_frames(321, 3, H, W)
    engine.preprocess(frames, _ffi.PH_PRE_NATIVE)
    engine.forward(3)
    got3 = engine.net_out(3)
    x = frames[1:2].astype(np.float32).transpose(0, 3, 1, 2) * np.float32(1 / 256.0) - np.float32(0.5)
    ref = net_ref.forward(net, weights, x)
    assert got3.shape == (3, 78, 46, 82) and np.isfinite(got3).all()
    err = np.abs(got3[1] - ref[0]).max()
    assert err < 1e-2, err                                   # heat-map max abs err (north_star)
    assert err / (np.abs(ref).max() + 1e-9) < 3e-2
    res3, _ = engine.decode_v2(BODY_25, 3, H, W, 1.0, 128)
    kp_ref, sc_ref = decode_v2.decode(got3[1], BODY_25, H, W, 1.0)
    np.testing.assert_array_equal(res3[1][0], kp_ref)
    np.testing.assert_array_equal(res3[1][1], sc_ref)
    engine.preprocess(frames[1:2], _ffi.PH_PRE_NATIVE)
    engine.forward(1)
    np.testing.assert_array_equal(engine.net_out(1)[0], got3[1])


def test_fused_pool_and_kernel_names(engine):
    """ph_op_desc.pool is validated (3x3, cin_read % 32 == 0, even H/W) and ph_op_kernel names what runs."""
    h, w, spec = CASES["vgg_like"]
    net = _tiny_net(h, w, spec)
    weights = graph.random_weights(net, seed=7, out_gain=1.0)
    plan = graph.build_plan(net, weights)
    assert [o.get("pool", 0) for o in plan.ops] == [0, 1, 0, 0]          # p1 folded into c2
    engine.build(plan, 1)
    names = engine.op_kernels()
    assert names[0] == "(fused into the next op)" and names[1].startswith("conv_stem512_kernel")   # c1 inside c2 (stem)
    assert names[3].endswith("1>") or names[3].startswith("conv_igemm_kernel")
    unfused = graph.build_plan(net, weights, fuse_pool=False)
    assert [o["type"] for o in unfused.ops].count(_ffi.PH_OP_POOL) == 1
    odd = _tiny_net(13, 19, [("conv", "c1", "image", 64, 3, "ReLU"), ("conv", "c2", "c1", 64, 3, "ReLU"),
                             ("pool", "p", "c2"), ("conv", "out", "p", 38, 3, None)])
    wodd = graph.random_weights(odd, seed=1)
    podd = graph.build_plan(odd, wodd)
    assert all(o.get("pool", 0) == 0 for o in podd.ops)                    # odd sizes: pooling stays a separate op
    podd.ops[1]["pool"] = 1                                                # forcing it must fail loudly
    with pytest.raises(_ffi.PosehipError):
        engine.build(podd, 1)


def test_body25_batch32_production_tiles_meet_the_oracle(engine):
    """The instantiations bench.py times (B = 32: 128- and 96-wide N tiles on the 46x82 stage layers, which
    smaller batches narrow to 32/64) against the oracle: frame k of the batch of 32 is bit-equal to the same
    frame run alone (different tile selection, same K order), and that run is within 1e-2 of the torch fp32
    oracle -- so the benched kernels meet the oracle transitively; two frames are also compared directly."""
    from oracle import net_ref
    from posehip import synth
    H, W, B = 368, 656, 32
    net = caffe_io.parse_prototxt(caffe_io.body25_prototxt(H, W))
    weights = graph.random_weights(net, seed=9)
    plan = graph.build_plan(net, weights)
    engine.build(plan, B)
    names = engine.op_kernels()
    assert names.count("conv3x3_halo_kernel<4, 4, false, false, 9>") == 70      # the dominant kernel of the bench
    assert names.count("conv3x3_halo_kernel<3, 4, false, false, 9>") == 30
    assert names.count("conv_stem512_kernel(PhConvParams, int, int, int)") == 1  # stem (conv1_1 + conv1_2 + pool)
    frames = synth.synth_frames(77, B, H, W)
    engine.preprocess(frames, _ffi.PH_PRE_NATIVE)
    engine.forward(B)
    got = engine.net_out(B)
    assert np.isfinite(got).all()
    for k in (0, 13, 31):
        x = frames[k:k + 1].astype(np.float32).transpose(0, 3, 1, 2) * np.float32(1 / 256.0) - np.float32(0.5)
        if k != 13:
            ref = net_ref.forward(net, weights, x)
            err = np.abs(got[k] - ref[0]).max()
            assert err < 1e-2, (k, err)
    engine.build(plan, 1)                         # single-frame tile selection (narrow N tiles)
    assert engine.op_kernels() != names
    for k in (0, 13, 31):
        engine.preprocess(frames[k:k + 1], _ffi.PH_PRE_NATIVE)
        engine.forward(1)
        np.testing.assert_array_equal(engine.net_out(1)[0], got[k])


@pytest.mark.parametrize("H,W,B", [(64, 96, 3), (80, 112, 2), (368, 656, 2)])
def test_persistent_stem_kernel_equals_the_halo_stem(engine, H, W, B, monkeypatch):
    """conv_stem512_kernel (persistent, weights resident in LDS, 16x32 tiles, pool inside the wave) against the halo
    kernel's stem form (POSEHIP_STEM_KERNEL=0): same accumulation order, so the pooled conv1_2 output and everything
    after it are bit-equal -- on sizes that are multiples of the tile, ragged in both directions, and full size."""
    from posehip import synth
    net = caffe_io.parse_prototxt(caffe_io.body25_prototxt(H, W))
    weights = graph.random_weights(net, seed=31)
    plan = graph.build_plan(net, weights)
    frames = synth.synth_frames(5, B, H, W)
    outs, pooled = [], []
    for flag in ("0", "1"):
        monkeypatch.setenv("POSEHIP_STEM_KERNEL", flag)
        engine.build(plan, B)
        names = engine.op_kernels()
        assert ("conv_stem512_kernel(PhConvParams, int, int, int)" in names) == (flag == "1")
        for _ in range(2):                      # twice: the tile counters alternate between launches
            engine.preprocess(frames, _ffi.PH_PRE_NATIVE)
            engine.forward(B)
        outs.append(engine.net_out(B))
        assert plan.ops[1].get("pool", 0) == 1
        pooled.append(engine.get_buffer(plan.ops[1]["dst"][0][0]))       # pool1: what the stem kernel writes
    assert np.isfinite(outs[0]).all() and np.abs(outs[0]).max() > 0
    np.testing.assert_array_equal(outs[0], outs[1])
    np.testing.assert_array_equal(pooled[0], pooled[1])
    assert np.abs(pooled[1]).max() > 0


PAIR_CASES = {
    # (h, w, cin of the wide layer, its width, its activation, narrow width, narrow activation)
    "stage0_l2": (20, 30, 288, 256, "PReLU", 52, None),      # 9 chunks x 2 slabs, ragged last tile
    "stage1_l1": (16, 48, 384, 512, "PReLU", 26, None),      # 12 chunks x 4 slabs
    "relu_full": (16, 16, 160, 128, "ReLU", 64, "PReLU"),    # one slab, every narrow channel, exactly one tile
    "coco_short": (23, 19, 128, 512, "ReLU", 38, None),      # 4 chunks per slab
    "no_act": (9, 11, 96, 128, None, 19, "ReLU"),            # 3 chunks, less than one tile
}


@pytest.mark.parametrize("case", sorted(PAIR_CASES))
@pytest.mark.parametrize("batch", [1, 3])
def test_pair_kernel_equals_the_separate_1x1_layers(engine, case, batch, monkeypatch):
    """conv1x1_pair_kernel (two consecutive 1x1 convs as one launch, the wide map only ever in LDS) against the two
    separate launches (POSEHIP_PAIR_KERNEL=0, read at every forward), one pair at a time: same operand roles, the
    intermediate rounded to bf16 and the narrow layer's K walked in the same ascending 32-chunks, so its fp32 output
    and a further 3x3 layer that reads its bf16 window are bit-equal."""
    h, w, cin, wide, act1, narrow, act2 = PAIR_CASES[case]
    spec = [("conv", "f", "image", cin, 3, "ReLU"), ("conv", "m6", "f", wide, 1, act1),
            ("conv", "m7", "m6", narrow, 1, act2), ("concat", "cat", ["f", "m7"]),
            ("conv", "tail", "cat", 24, 3, None), ("concat", "net_output", ["m7", "tail"])]
    net = _tiny_net(h, w, spec)
    weights = graph.random_weights(net, seed=23, out_gain=1.0)
    plan = graph.build_plan(net, weights)
    engine.build(plan, batch)
    x = np.random.default_rng(4).integers(0, 256, (batch, h, w, 3), dtype=np.uint8)
    outs = []
    for flag in ("0", "1"):
        monkeypatch.setenv("POSEHIP_PAIR_KERNEL", flag)
        names = engine.op_kernels()
        assert ("conv1x1_pair_kernel(PhPairParams)" in names) == (flag == "1"), names
        engine.preprocess(x, _ffi.PH_PRE_CV2)
        engine.forward(batch)
        outs.append(engine.net_out(batch))
    assert np.isfinite(outs[1]).all()
    assert np.abs(outs[0][:, :narrow]).max() > 0 and np.abs(outs[0][:, narrow:]).max() > 0
    np.testing.assert_array_equal(outs[0], outs[1])
    engine.preprocess(x[-1:], _ffi.PH_PRE_CV2)      # a frame alone: another tile count, another place in its tile
    engine.forward(1)
    np.testing.assert_array_equal(engine.net_out(1)[0], outs[1][-1])


@pytest.mark.parametrize("model,H,W,B", [("body25", 64, 96, 3), ("body25", 368, 656, 2), ("coco", 80, 112, 2)])
def test_pair_kernel_in_the_full_graphs(engine, model, H, W, B, monkeypatch):
    """The pair kernel inside BODY_25 / COCO-18: every stage's Mconv6 + Mconv7 (COCO: also conv5_4 + conv5_5) runs on
    it, and the network output equals the separate launches' bit for bit -- tiles ragged at the end and full size, a
    smaller batch in between (the tile count follows it)."""
    from posehip import synth
    proto = caffe_io.body25_prototxt(H, W) if model == "body25" else caffe_io.coco_prototxt(H, W)
    net = caffe_io.parse_prototxt(proto)
    weights = graph.random_weights(net, seed=17)
    plan = graph.build_plan(net, weights)
    engine.build(plan, B)
    frames = synth.synth_frames(9, B, H, W)
    outs = []
    for flag in ("0", "1"):
        monkeypatch.setenv("POSEHIP_PAIR_KERNEL", flag)
        names = engine.op_kernels()
        pair_ops = [i for i, k in enumerate(names) if k == "conv1x1_pair_kernel(PhPairParams)"]
        if flag == "0":
            assert not pair_ops
        else:
            assert len(pair_ops) >= 6 and all(names[i - 1] == "(fused into the next op)" for i in pair_ops)
        for n in (B, 1, B):
            engine.preprocess(frames[:n], _ffi.PH_PRE_NATIVE)
            engine.forward(n)
        outs.append(engine.net_out(B))
    assert np.isfinite(outs[0]).all() and np.abs(outs[0]).max() > 0
    np.testing.assert_array_equal(outs[0], outs[1])


def test_coco_full_size_forward_and_v1_decode(engine):
    """BASELINE configs[2] geometry: the COCO-18 graph (7x7 refinement stages on the im2col kernel) at 368x656,
    B = 2, against the torch fp32 oracle; then the V1 decode (postprocess_multi semantics, 1920x1080 target) of the
    GPU's OWN network output against the oracle's decode of that output, bit for bit."""
    from oracle import decode_v1, net_ref
    from posehip import synth, tables
    H, W = 368, 656
    net = caffe_io.parse_prototxt(caffe_io.coco_prototxt(H, W))
    weights = graph.random_weights(net, seed=13)
    plan = graph.build_plan(net, weights)
    engine.build(plan, 2)
    assert any(k.startswith("conv_igemm_kernel") for k in engine.op_kernels())      # the 7x7 layers
    frames = synth.synth_frames(55, 2, H, W)
    engine.preprocess(frames, _ffi.PH_PRE_CV2)
    engine.forward(2)
    got = engine.net_out(2)
    assert got.shape == (2, 57, 46, 82) and np.isfinite(got).all()
    x = frames[1:2].astype(np.float32).transpose(0, 3, 1, 2) * np.float32(1 / 255.0)
    ref = net_ref.forward(net, weights, x)
    err = np.abs(got[1] - ref[0]).max()
    assert err < 1e-2, err
    assert err / (np.abs(ref).max() + 1e-9) < 3e-2
    bp, pp = tables.get_body_parts_and_pose_pairs("COCO")
    # random weights give noise maps: a threshold at their 99.7th percentile leaves a few dozen blobs per part
    # (the default 0.1 would exceed the 8192 mask runs per part the decode is sized for)
    thr = float(np.quantile(got[1, :18], 0.997))
    res = engine.decode_v1(bp, pp, tables.MAPIDX, 2, 1920, 1080, thr)
    kl_o, people_o = decode_v1.postprocess_multi(got[1:2], bp, pp, tables.MAPIDX, 1920, 1080, thr)
    assert len(kl_o) > 18
    np.testing.assert_array_equal(res[1][0], kl_o)
    assert len(res[1][1]) == len(people_o)
    for a, b in zip(res[1][1], people_o):
        np.testing.assert_array_equal(a, b)


def test_body25_single_frame_368x368(engine):
    """BASELINE configs[0] geometry: one 368x368 frame through BODY_25 (single-frame tile selection) against the
    oracle, and its V2 decode against the oracle's decode of the same output."""
    from oracle import decode_v2, net_ref
    from posehip import synth
    from posehip.tables import BODY_25
    H = W = 368
    net = caffe_io.parse_prototxt(caffe_io.body25_prototxt(H, W))
    weights = graph.random_weights(net, seed=17)
    engine.build(graph.build_plan(net, weights), 1)
    frame = synth.synth_frames(99, 1, H, W)
    engine.preprocess(frame, _ffi.PH_PRE_NATIVE)
    engine.forward(1)
    got = engine.net_out(1)
    x = frame.astype(np.float32).transpose(0, 3, 1, 2) * np.float32(1 / 256.0) - np.float32(0.5)
    ref = net_ref.forward(net, weights, x)
    assert got.shape == (1, 78, 46, 46)
    assert np.abs(got - ref).max() < 1e-2
    res, _ = engine.decode_v2(BODY_25, 1, H, W, 1.0, 128)
    kp_o, sc_o = decode_v2.decode(got[0], BODY_25, H, W, 1.0)
    np.testing.assert_array_equal(res[0][0], kp_o)
    np.testing.assert_array_equal(res[0][1], sc_o)
