"""GPU parity: fused upsample+NMS / PAF matching / assembly kernels vs the V2
oracle (oracle/decode_v2.py), through the C ABI.  Integer outputs (peak counts,
subset indices) and the fp32 values are bit-exact by construction (same IEEE
operation order, -ffp-contract=off)."""
import numpy as np
import pytest

from posehip import _ffi, synth
from posehip.tables import BODY_25, COCO_18

pytestmark = pytest.mark.gpu


_VARIANTS = {}


def _m(model, connector):
    """The model with upstream's CPU-build (0) or CUDA-build (1) person assembly."""
    key = (model.name, connector)
    if key not in _VARIANTS:
        _VARIANTS[key] = model.with_connector(connector)
    return _VARIANTS[key]


def _oracle(net_out, model, Ht, Wt, scale):
    from oracle import decode_v2
    return decode_v2.decode(net_out, model, Ht, Wt, scale, return_intermediates=True)


def _check_frame(kp_o, sc_o, inter, got, extras, b, model, engine=None):
    kp_g, sc_g = got[b]
    np.testing.assert_array_equal(extras["peaks"][b], inter["peaks"])
    if engine is not None and model.connector == 0:     # per-limb connection lists exist on the CPU-build route only
        n, idx, cs = engine.decode_v2_connections(model, len(got))
        for k, conns in enumerate(inter["connections"]):
            assert n[b, k] == len(conns), (k, n[b, k], len(conns))
            for c, (ia, ib, s) in enumerate(conns):
                assert (idx[b, k, c, 0], idx[b, k, c, 1]) == (ia, ib)
                assert cs[b, k, c].view(np.uint32) == np.float32(s).view(np.uint32), (k, c, cs[b, k, c], s)
    assert int(extras["n_people"][b]) == len(inter["valid"])
    rows = inter["rows"][inter["valid"]] if len(inter["valid"]) else np.zeros((0, model.n_parts + 1), np.int32)
    np.testing.assert_array_equal(extras["subsets"][b, :len(rows)], rows)
    np.testing.assert_array_equal(kp_g, kp_o)
    np.testing.assert_array_equal(sc_g, sc_o)


@pytest.mark.parametrize("model,hs,ws,people", [
    (BODY_25, 46, 82, [0, 1, 2, 5]),
    (BODY_25, 46, 46, [3]),
    (COCO_18, 46, 82, [4]),
    (BODY_25, 23, 31, [2]),
])
@pytest.mark.parametrize("connector", [0, 1])
def test_decode_matches_oracle(engine, model, hs, ws, people, connector):
    model = _m(model, connector)
    frames = np.stack([synth.synth_net_output(100 + i, p, model, hs, ws) for i, p in enumerate(people)])
    engine.set_net_out(frames)
    Ht, Wt = hs * 8, ws * 8
    got, extras = engine.decode_v2(model, len(people), Ht, Wt, scale=1.75, want_peaks=True, want_subsets=True)
    found = 0
    for b in range(len(people)):
        kp_o, sc_o, inter = _oracle(frames[b], model, Ht, Wt, 1.75)
        _check_frame(kp_o, sc_o, inter, got, extras, b, model, engine)
        found += len(sc_o)
    if sum(people) > 0:
        assert found > 0                         # the scenes do produce skeletons


@pytest.mark.parametrize("connector", [0, 1])
def test_crowded_and_truncation(engine, connector):
    BODY_25 = _m(globals()["BODY_25"], connector)
    # 20 people: stresses ordered compaction; plus a noise frame that overflows 127 peaks/part
    rng = np.random.default_rng(5)
    crowded = synth.synth_net_output(7, 20, BODY_25, 46, 82)
    noise = rng.uniform(0.0, 1.0, crowded.shape).astype(np.float32)
    noise[BODY_25.n_heat:] = rng.normal(0, 0.3, noise[BODY_25.n_heat:].shape)
    frames = np.stack([crowded, noise])
    engine.set_net_out(frames)
    got, extras = engine.decode_v2(BODY_25, 2, 368, 656, want_peaks=True, want_subsets=True, max_people=512)
    assert (extras["peaks"][1, :, 0, 0] == 127).all()       # truncated to max_peaks in raster order
    for b in range(2):
        kp_o, sc_o, inter = _oracle(frames[b], BODY_25, 368, 656, 1.0)
        _check_frame(kp_o, sc_o, inter, got, extras, b, BODY_25, engine)


@pytest.mark.parametrize("hs,ws", [
    (12, 130),    # 96 x 1040 target: two column groups per wave in the NMS kernel
    (14, 260),    # 112 x 2080 target: four column groups per wave
    (92, 164),    # 736 x 1312 target: PAF planes too large for coefficient tables -> raw planes in LDS
])
@pytest.mark.parametrize("connector", [0, 1])
def test_wide_targets(engine, hs, ws, connector):
    BODY_25 = _m(globals()["BODY_25"], connector)
    f = synth.synth_net_output(11, 3, BODY_25, hs, ws)[None]
    engine.set_net_out(f)
    got, extras = engine.decode_v2(BODY_25, 1, hs * 8, ws * 8, want_peaks=True, want_subsets=True)
    kp_o, sc_o, inter = _oracle(f[0], BODY_25, hs * 8, ws * 8, 1.0)
    _check_frame(kp_o, sc_o, inter, got, extras, 0, BODY_25, engine)
    assert extras["peaks"][0, :, 0, 0].sum() > 0


@pytest.mark.parametrize("seed", range(6))
@pytest.mark.parametrize("connector", [0, 1])
def test_random_sizes_and_ratios(engine, seed, connector):
    BODY_25 = _m(globals()["BODY_25"], connector)
    # non-integer upsampling ratios (runs of rows with equal vertical taps of uneven length), odd sizes, stripes that
    # end inside a band, widths around the 62-column groups of the NMS kernel
    rng = np.random.default_rng(900 + seed)
    hs, ws = int(rng.integers(14, 40)), int(rng.integers(14, 60))
    Ht = int(hs * rng.uniform(4.0, 9.0)) + int(rng.integers(0, 3))
    Wt = int(ws * rng.uniform(4.0, 9.0)) + int(rng.integers(0, 3))
    Ht, Wt = max(Ht, 4 * hs), max(Wt, 4 * ws)
    f = np.stack([synth.synth_net_output(50 + seed, 2, BODY_25, hs, ws),
                  synth.synth_net_output(70 + seed, 4, BODY_25, hs, ws)])
    engine.set_net_out(f)
    got, extras = engine.decode_v2(BODY_25, 2, Ht, Wt, scale=1.25, want_peaks=True, want_subsets=True)
    for b in range(2):
        kp_o, sc_o, inter = _oracle(f[b], BODY_25, Ht, Wt, 1.25)
        _check_frame(kp_o, sc_o, inter, got, extras, b, BODY_25, engine)


def test_non_finite_heat_maps_take_the_nan_aware_rows(engine):
    # nms_detect_cols_kernel walks planes whose values are all finite and < 1e30 without NaN bookkeeping; a plane with
    # a NaN, an infinity or a value whose cubic overflows takes the other loop body, where `v > NaN` is false in each
    # of the eight neighbour tests as in the oracle's array comparisons.  Other planes of the same frame stay fast.
    from oracle import decode_v2
    f = np.stack([synth.synth_net_output(21, 3, BODY_25, 46, 82), synth.synth_net_output(22, 2, BODY_25, 46, 82)])
    f[0, 0, 10, 20] = np.nan
    f[0, 1, 5, 5] = np.inf
    f[0, 2, 30, 60] = -np.inf
    f[0, 3, 20, 40] = 3e38          # finite, but the cubic of it overflows: inf - inf further on
    f[0, 4, 0, 0] = np.nan          # corner: clamped taps spread it over the border rows
    f[1, 7, 45, 81] = 2e30          # finite and harmless, still above the fast path's bound
    engine.set_net_out(f)
    got, extras = engine.decode_v2(BODY_25, 2, 368, 656, want_peaks=True)
    for b in range(2):
        with np.errstate(all="ignore"):
            up = decode_v2.upsample_bicubic(f[b, :BODY_25.n_parts], 368, 656)
            want = decode_v2.nms(up, BODY_25.nms_threshold, BODY_25.max_peaks)
        np.testing.assert_array_equal(extras["peaks"][b, :, 0, 0], want[:, 0, 0])
        for p in range(BODY_25.n_parts):
            n = int(want[p, 0, 0])
            # integer peak locations and scores (the refined centroid of a peak next to a NaN is a NaN on both sides)
            np.testing.assert_array_equal(extras["peaks"][b, p, 1:n + 1, 2].view(np.uint32),
                                          want[p, 1:n + 1, 2].view(np.uint32))
            ok = ~np.isnan(want[p, 1:n + 1, 0])
            np.testing.assert_array_equal(extras["peaks"][b, p, 1:n + 1, :2][ok], want[p, 1:n + 1, :2][ok])
    assert extras["peaks"][:, :, 0, 0].sum() > 0


def test_empty_and_flat_maps(engine):
    z = np.zeros((2, BODY_25.n_channels, 46, 82), np.float32)
    z[1] = 0.5                                   # plateau: strict > never fires
    engine.set_net_out(z)
    got, extras = engine.decode_v2(BODY_25, 2, 368, 656, want_peaks=True)
    assert (extras["n_people"] == 0).all()
    assert (extras["peaks"] == 0).all()
    assert got[0][0].shape == (0, 25, 3) and got[0][1].shape == (0,)


@pytest.mark.parametrize("connector", [0, 1])
def test_non_x8_target_and_errors(engine, connector):
    from posehip._ffi import PosehipError
    BODY_25 = _m(globals()["BODY_25"], connector)
    f = synth.synth_net_output(3, 2, BODY_25, 46, 82)[None]
    engine.set_net_out(f)
    got, extras = engine.decode_v2(BODY_25, 1, 46 * 6, 82 * 5, want_peaks=True, want_subsets=True)
    kp_o, sc_o, inter = _oracle(f[0], BODY_25, 46 * 6, 82 * 5, 1.0)
    _check_frame(kp_o, sc_o, inter, got, extras, 0, BODY_25, engine)
    with pytest.raises(PosehipError):
        engine.decode_v2(BODY_25, 1, 46, 82)     # below the supported upsampling ratio
    engine.set_net_out(np.zeros((1, 30, 46, 82), np.float32))
    with pytest.raises(PosehipError):
        engine.decode_v2(BODY_25, 1, 368, 656)   # too few channels for BODY_25


def test_pose_post_matches_oracle_bit_exact(engine):
    """torso boxes (fp32) and depth lift (fp64) computed on the device keypoints of the last decode
    equal oracle/post_v2.py bit for bit, for host and device depth images."""
    import torch
    from oracle import post_v2
    from posehip import synth
    from posehip.tables import BODY_25, TORSO_IDS
    B, hs, ws = 3, 46, 82
    net = np.stack([synth.synth_net_output(900 + i, [4, 0, 7][i], BODY_25, hs, ws) for i in range(B)])
    engine.set_net_out(net)
    res, ex = engine.decode_v2(BODY_25, B, hs * 8, ws * 8, 1080 / 368, 128)
    rng = np.random.default_rng(5)
    depth = rng.integers(300, 6000, (B, 1080, 1920), dtype=np.uint16)
    intr = (910.5, 911.25, 958.7, 541.3)
    for dev_depth in (False, True):
        d = torch.from_numpy(depth.view(np.int16)).cuda() if dev_depth else depth
        bbox, kp3d = engine.pose_post(B, 128, BODY_25.n_parts, TORSO_IDS, d, intr, 1e-3, 2)
        for b in range(B):
            kp = res[b][0]
            m = len(kp)
            assert m == [4, 0, 7][b] or m > 0 or b == 1
            np.testing.assert_array_equal(bbox[b, :m], post_v2.torso_bbox(kp, TORSO_IDS))
            assert (bbox[b, m:] == 0).all() and (kp3d[b, m:] == 0).all()
            for p in range(m):
                want = post_v2.lift_3d(kp[p], depth[b], *intr, 1e-3, 2)
                np.testing.assert_array_equal(kp3d[b, p].view(np.uint64), want.view(np.uint64))
    with pytest.raises(_ffi.PosehipError):
        engine.pose_post(B, 128, BODY_25.n_parts, [99], None)             # torso id out of range


def test_exported_heatmaps_follow_heatmaps_scale(engine):
    """datum.poseHeatMaps export: bicubic maps mapped by --heatmaps_scale, bit-exact vs oracle/features.py."""
    from oracle import features
    hs, ws = 12, 20
    rng = np.random.default_rng(31)
    net = rng.normal(0.3, 0.8, (2, BODY_25.n_channels, hs, ws)).astype(np.float32)   # values beyond [-1,1] too
    engine.set_net_out(net)
    for mode in (0, 1, 2, 3):
        for c0, nc in ((0, 25), (25, 1), (26, 52), (20, 12)):
            got = engine.upsample_maps(1, c0, nc, hs * 8, ws * 8, BODY_25.n_heat, mode)
            want = features.export_heatmaps(net[1], [(c0, nc)], BODY_25.n_heat, mode, hs * 8, ws * 8)
            np.testing.assert_array_equal(got.view(np.uint32), want.view(np.uint32), err_msg=f"mode {mode} window {c0}")
    with pytest.raises(_ffi.PosehipError):
        engine.upsample_maps(0, 0, 25, hs * 8, ws * 8, BODY_25.n_heat, 4)


def test_pose_features_match_oracle_bit_exact(engine):
    """the tracker feature (create_detections' crop of the exported heat maps under the torso box +
    resize_tensor to 25x50, tracking/utils_track.py:45-57) computed per person from the resident net output equals
    the oracle's crop of the fully materialised maps, bit for bit -- all scale modes, channel selections, an image
    size different from the net input size, frames without people, degenerate boxes."""
    from oracle import features, post_v2
    from posehip.tables import TORSO_IDS
    B, hs, ws = 3, 46, 82
    Ht, Wt = hs * 8, ws * 8
    net = np.stack([synth.synth_net_output(700 + i, [3, 0, 5][i], BODY_25, hs, ws) for i in range(B)])
    engine.set_net_out(net)
    all_win = [(0, 25), (25, 1), (26, 52)]
    for image_size, scale, mode, windows, size in (((656, 368), 1.0, 1, all_win, (25, 50)),
                                                   ((1920, 1080), 1080 / 368, 2, [(0, 25)], (25, 50)),
                                                   ((1920, 1080), 1080 / 368, 0, [(0, 25), (26, 52)], (7, 4)),
                                                   ((656, 368), 1.0, 3, [(25, 1)], (25, 50))):
        res, ex = engine.decode_v2(BODY_25, B, Ht, Wt, scale, 128)
        counts = [len(res[b][1]) for b in range(B)]
        assert counts[1] == 0 and counts[0] > 0 and counts[2] > 0
        feat, status = engine.pose_features(B, 128, BODY_25.n_parts, TORSO_IDS, windows, BODY_25.n_heat, mode, Ht, Wt,
                                            image_size, sum(counts), size)
        nch = sum(nc for _, nc in windows)
        assert feat.shape == (sum(counts), size[1], size[0], nch) and status.shape == (sum(counts),)
        row = 0
        for b in range(B):
            if not counts[b]:
                continue
            hm = np.moveaxis(features.export_heatmaps(net[b], windows, BODY_25.n_heat, mode, Ht, Wt), 0, -1)
            want = features.detection_features(post_v2.torso_bbox(res[b][0], TORSO_IDS), hm, image_size, size)
            for w in want:
                if isinstance(w, Exception):
                    assert status[row] == -1 and not feat[row].any()
                else:
                    assert status[row] == 1
                    assert w.dtype == np.float64
                    np.testing.assert_array_equal(feat[row].reshape(-1).astype(np.float64), w)
                    assert feat[row].any()
                row += 1
        assert row == sum(counts)
    # a buffer too small for the people found is refused, not overrun
    with pytest.raises(_ffi.PosehipError):
        engine.pose_features(B, 128, BODY_25.n_parts, TORSO_IDS, all_win, BODY_25.n_heat, 1, Ht, Wt, (656, 368), 2)
    with pytest.raises(_ffi.PosehipError):
        engine.pose_features(B, 128, BODY_25.n_parts, TORSO_IDS, [(70, 20)], BODY_25.n_heat, 1, Ht, Wt, (656, 368), 8)


def test_pose_features_degenerate_crop_flags_index_error(engine):
    """A person whose torso box maps to a one-column crop: the reference's resize_tensor raises IndexError; the
    kernel reports status -1 and zeros."""
    from posehip.tables import TORSO_IDS
    hs, ws = 46, 82
    net = synth.synth_net_output(701, 2, BODY_25, hs, ws)[None]
    engine.set_net_out(net)
    res, _ = engine.decode_v2(BODY_25, 1, hs * 8, ws * 8, 1.0, 128)
    n = len(res[0][1])
    assert n > 0
    # an image "size" so wide that every box collapses to <= 1 heat-map column
    feat, status = engine.pose_features(1, 128, BODY_25.n_parts, TORSO_IDS, [(0, 25)], BODY_25.n_heat, 1, hs * 8, ws * 8,
                                        (656 * 4000, 368), n)
    assert (status == -1).all() and not feat.any()


def test_person_table_overflow_is_reported_on_both_paths(engine):
    """More than 2048 partial skeletons in a frame: ph_decode_v2 returns -2, and so does ph_decode_v2_status after
    the asynchronous ph_decode_v2_dev + ph_decode_v2_results (the path BatchPipeline and bench.py use)."""
    import torch
    from posehip.tables import PoseModelV2
    # star model: 23 limbs (k, 0) whose B side (part 0) has no peak -> every peak of the 23 other parts becomes a
    # single-part person: 23 x 127 = 2921 rows (upstream createPeopleVector, the "one side empty" branch)
    star = PoseModelV2("star", [f"p{i}" for i in range(24)] + ["Background"], [(k, 0) for k in range(1, 24)],
                       [(2 * k, 2 * k + 1) for k in range(23)], (-1, -1), connector=0)
    rng = np.random.default_rng(3)
    f = np.zeros((1, star.n_channels, 46, 82), np.float32)
    f[0, 1:24] = rng.uniform(0.1, 1.0, (23, 46, 82)).astype(np.float32)
    engine.set_net_out(f)
    with pytest.raises(_ffi.PosehipError, match=r"\(-2\).*overflow"):
        engine.decode_v2(star, 1, 368, 656, max_people=128)
    pn, pk, ps = engine.decode_v2_dev(star, 1, 368, 656, max_people=128)
    n = torch.zeros((1,), dtype=torch.int32).pin_memory()
    kp = torch.zeros((1, 128, 24, 3), dtype=torch.float32).pin_memory()
    sc = torch.zeros((1, 128), dtype=torch.float32).pin_memory()
    engine.decode_v2_results(n, kp, sc)
    with pytest.raises(_ffi.PosehipError, match=r"\(-2\).*overflow"):
        engine.decode_v2_status()
    # a frame that fits clears the flag again
    ok = synth.synth_net_output(5, 2, BODY_25, 46, 82)[None]
    engine.set_net_out(ok)
    engine.decode_v2_dev(BODY_25, 1, 368, 656)
    kp25 = torch.zeros((1, 128, 25, 3), dtype=torch.float32).pin_memory()
    engine.decode_v2_results(n, kp25, sc)
    engine.decode_v2_status()
    assert int(n[0]) >= 1


def test_cuda_connector_merge_scene_and_matrix_mode(engine):
    """connector 1 on cases built to hit its branches: (a) two partial skeletons of one person that only a late limb
    joins (merge-if-disjoint), (b) more accepted candidate pairs than the connection list holds, so the rounds start
    on the score matrix (noise PAFs: ~10^5 accepted pairs), (c) exact score ties (identical people)."""
    from oracle import decode_v2
    m1 = _m(BODY_25, 1)
    # (a) a person whose mid-hip -> right-hip PAF is weak: upper body and right leg are founded separately and the
    # (8, 9) connection, which comes late in the score order, merges them
    base = synth.synth_net_output(41, 3, BODY_25, 46, 82)
    weak = base.copy()
    k = BODY_25.pairs.index((8, 9))
    for ch in BODY_25.mapidx[k]:
        weak[BODY_25.n_heat + ch] *= 0.35
    # (b) uniform-noise heat maps + PAFs that accept most pairs
    rng = np.random.default_rng(8)
    noise = rng.uniform(0.0, 1.0, base.shape).astype(np.float32)
    noise[BODY_25.n_heat:] = rng.normal(0.35, 0.25, noise[BODY_25.n_heat:].shape)
    # (c) the same person twice, shifted by whole cells: identical peak scores and PAF sums
    one = synth.synth_net_output(43, 1, BODY_25, 46, 40, noise=0.0)
    twins = np.concatenate([one, one], axis=2)[:, :, :82]
    twins = np.concatenate([twins, np.zeros((twins.shape[0], 46, 82 - twins.shape[2]), np.float32)], axis=2)
    frames = np.stack([base, weak, noise, twins])
    engine.set_net_out(frames)
    got, extras = engine.decode_v2(m1, 4, 368, 656, scale=1.0, want_peaks=True, want_subsets=True, max_people=512)
    merged = 0
    for b in range(4):
        kp_o, sc_o, inter = decode_v2.decode(frames[b], m1, 368, 656, 1.0, return_intermediates=True)
        _check_frame(kp_o, sc_o, inter, got, extras, b, m1)
        if b == 2:
            assert len(inter["connections"]) > 16384          # the matrix-scan rounds ran
        merged += len(inter["connections"])
    assert len(got[1][1]) >= 1
