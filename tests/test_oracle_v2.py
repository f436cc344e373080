"""CPU: the V2 oracle against known-answer cases, the committed golden scenes and
the one result artefact the reference holds (predictions.txt)."""
import os

import numpy as np
import pytest

from oracle import decode_v2
from posehip.tables import BODY_25, COCO_18

G = os.path.join(os.path.dirname(__file__), "golden")


def test_bicubic_kats():
    c = np.full((3, 5, 7), 0.25, np.float32)
    up = decode_v2.upsample_bicubic(c, 40, 56)
    assert np.allclose(up, 0.25, atol=1e-6)                     # partition of unity
    ramp = np.tile(np.arange(8, dtype=np.float32), (6, 1))
    up = decode_v2.upsample_bicubic(ramp, 48, 64)
    xs = (np.arange(64) + 0.5) / 8 - 0.5
    inner = (xs >= 1) & (xs <= 6)
    assert np.allclose(up[10, inner], xs[inner], atol=1e-5)     # Catmull-Rom reproduces linear ramps
    # single pixel == full map (the kernels sample single pixels)
    rng = np.random.default_rng(0)
    m = rng.normal(size=(1, 9, 11)).astype(np.float32)
    full = decode_v2.upsample_bicubic(m, 72, 88)
    x0, x1, x2, x3, dx = decode_v2._src_coords(88, 11)
    y0, y1, y2, y3, dy = decode_v2._src_coords(72, 9)
    for (y, x) in [(0, 0), (71, 87), (35, 40), (3, 86)]:
        t = [decode_v2._cubic(m[0, r, x0[x]], m[0, r, x1[x]], m[0, r, x2[x]], m[0, r, x3[x]], dx[x])
             for r in (y0[y], y1[y], y2[y], y3[y])]
        assert decode_v2._cubic(t[0], t[1], t[2], t[3], dy[y]) == full[0, y, x]


def test_nms_kats():
    up = np.zeros((1, 20, 30), np.float32)
    up[0, 5, 5] = 0.9            # isolated peak
    up[0, 0, 10] = 0.9           # on the border: excluded
    up[0, 10, 10] = up[0, 10, 11] = 0.5    # plateau: strict > never fires
    up[0, 15, 20] = 0.04         # below threshold
    up[0, 12, 25] = 0.3
    pk = decode_v2.nms(up, 0.05, 127)
    assert pk[0, 0, 0] == 2
    assert tuple(pk[0, 1]) == (5.5, 5.5, np.float32(0.9))       # +0.5 offset, raster order
    assert tuple(pk[0, 2]) == (25.5, 12.5, np.float32(0.3))
    # truncation keeps the first max_peaks in raster order
    many = np.zeros((1, 40, 40), np.float32)
    many[0, 4:36:8, 4:36:8] = np.linspace(0.1, 1.0, 16).reshape(4, 4)
    pk = decode_v2.nms(many, 0.05, 10)
    assert pk[0, 0, 0] == 10
    assert list(pk[0, 1:, 1]) == [4.5] * 4 + [12.5] * 4 + [20.5] * 2    # first 10 of 16 in raster order
    # 7x7 centroid pulls towards a heavier neighbour
    two = np.zeros((1, 20, 20), np.float32)
    two[0, 10, 10], two[0, 10, 12] = 1.0, 0.5
    pk = decode_v2.nms(two, 0.05, 127)
    assert pk[0, 0, 0] == 2 and 10.5 < pk[0, 1, 0] < 11.5


def test_pair_scores_vector_equals_scalar_and_rules():
    rng = np.random.default_rng(3)
    mx = rng.normal(0, 0.5, (64, 96)).astype(np.float32)
    my = rng.normal(0, 0.5, (64, 96)).astype(np.float32)
    mx[:, :] += 0.8                                            # mostly aligned with +x
    a = np.array([[5.5, 10.5, .9], [20.5, 30.5, .8], [50.5, 5.5, .7], [50.5, 5.5, .7]], np.float32)
    b = np.array([[80.5, 12.5, .9], [22.0, 31.0, .8], [50.5, 5.5, .6]], np.float32)
    v = decode_v2.pair_scores(a, b, mx, my, 0.05, 0.95, 0.05)
    for i in range(len(a)):
        for j in range(len(b)):
            s = decode_v2.score_ab(a[i], b[j], mx, my, 0.05, 0.95, 0.05)
            assert s.view(np.uint32) == v[i, j].view(np.uint32)
    assert v[2, 2] == 0 and v[3, 2] == 0                       # coincident peaks never connect
    # close-keypoint rescue: distance < sqrt(W*H)/150 scores nms_threshold + 1e-6
    z = np.zeros((368, 656), np.float32)
    s = decode_v2.score_ab(np.float32([10, 10, 1]), np.float32([12, 10, 1]), z, z, 0.05, 0.95, 0.05)
    assert s == np.float32(0.05 + 1e-6)
    s = decode_v2.score_ab(np.float32([10, 10, 1]), np.float32([20, 10, 1]), z, z, 0.05, 0.95, 0.05)
    assert s == 0


def test_sorted_greedy_and_assembly_rules():
    model = BODY_25
    P1 = model.max_peaks + 1
    peaks = np.zeros((25, P1, 3), np.float32)
    # limb 0 = (1, 8): two necks, two mid-hips; PAF favours neck0-hip1 and neck1-hip1 -> conflict on hip1
    peaks[1, 0, 0] = 2
    peaks[1, 1] = (100, 100, .9)
    peaks[1, 2] = (300, 100, .8)
    peaks[8, 0, 0] = 2
    peaks[8, 1] = (100, 200, .7)
    peaks[8, 2] = (200, 200, .6)
    H, W = 368, 656
    maps = np.zeros((78, H, W), np.float32)
    ch = model.n_heat + model.mapidx[0][0]
    maps[ch + 1] = 1.0                                         # PAF points straight down (+y)
    conns = decode_v2.select_connections(peaks, maps, model, 0)
    off = 3 * P1
    # scores: neck0-hip0 = 1 (vertical), neck0-hip1 = cos(45deg), neck1-hip1 = cos(45deg), neck1-hip0 lower
    assert [(c[0], c[1]) for c in conns] == [(1 * off + 1 * 3 + 2, 8 * off + 1 * 3 + 2),
                                             (1 * off + 2 * 3 + 2, 8 * off + 2 * 3 + 2)]
    all_conns = [[] for _ in range(model.n_pairs)]
    all_conns[0] = conns
    rows, scores = decode_v2.assemble(peaks, all_conns, model)
    assert rows.shape == (2, 26) and (rows[:, -1] == 2).all()
    assert scores[0] == np.float32(np.float32(np.float32(.9) + np.float32(.7)) + conns[0][2])
    valid = decode_v2.filter_people(rows, scores, model)
    assert valid == []              # 2 parts < minSubsetCnt (3), also on the maximizePositives retry
    kp, sc = decode_v2.people_array(rows, scores, [0, 1], peaks, model, 2.0)
    assert kp.shape == (2, 25, 3) and tuple(kp[0, 1]) == (200.0, 200.0, np.float32(.9)) and (kp[0, 0] == 0).all()


def test_filter_rules():
    model = BODY_25
    rows = np.zeros((3, 26), np.int32)
    rows[0, [1, 8, 9]] = 5
    rows[0, -1] = 3                                            # 3 body parts, good score -> kept
    rows[1, [1, 8, 19, 20]] = 5
    rows[1, -1] = 4                                            # 2 body + 2 foot parts -> counter 2 -> dropped
    rows[2, [1, 8, 9]] = 5
    rows[2, -1] = 3                                            # low mean score -> dropped
    scores = np.float32([3 * 0.5, 4 * 0.9, 3 * 0.3])
    assert decode_v2.filter_people(rows, scores, model) == [0]
    # nobody passes -> retry without the foot rule
    assert decode_v2.filter_people(rows[1:2], scores[1:2], model) == [0]


def test_reference_predictions_file_format():
    """The only stored output of the reference's native path pins the FORMAT of
    (poseScore, 25 x (x, y, score)) rows (native/python/skeleton.py:186-199) and
    the score convention (sum of part + limb scores) / (25 + 26)."""
    row = np.loadtxt(os.path.join(G, "ref_predictions_native.txt"), delimiter=",").reshape(1, -1)
    assert row.shape == (1, 76)
    score, kp = row[0, 0], row[0, 1:].reshape(25, 3)
    assert (kp[18] == 0).all() and (kp[np.arange(25) != 18, 2] > 0.05).all()   # undetected joint = zero triplet
    assert (kp[:, 0] <= 384).all() and (kp[:, 1] <= 384).all()                  # 384x384 frame (test.py:35)
    limb_sum = score * (BODY_25.n_parts + BODY_25.n_pairs) - kp[:, 2].sum()
    present = [k for k, (a, b) in enumerate(BODY_25.pairs) if kp[a, 2] > 0 and kp[b, 2] > 0]
    assert len(present) == 24                                                   # both LEar limbs missing
    assert 0.05 < limb_sum / len(present) <= 1.0                                # mean limb score is a PAF score
    # our packer writes the same layout
    rows = np.zeros((1, 26), np.int32)
    assert decode_v2.people_array(rows, np.float32([1.0]), [0], np.zeros((25, 128, 3), np.float32),
                                  BODY_25)[0].shape == (1, 25, 3)


@pytest.mark.parametrize("tag,model", [("b25_p3", BODY_25), ("b25_p0", BODY_25), ("b25_sq_p2", BODY_25),
                                       ("coco_p4", COCO_18)])
@pytest.mark.parametrize("connector", [0, 1])
def test_golden_scenes(tag, model, connector):
    g = np.load(os.path.join(G, "v2_scenes.npz"))
    sfx = "1" if connector else ""
    net = g[tag + "_net"].astype(np.float32)
    kp, sc, inter = decode_v2.decode(net, model, net.shape[1] * 8, net.shape[2] * 8, 1.0, return_intermediates=True,
                                     connector=connector)
    np.testing.assert_array_equal(kp, g[tag + "_kp" + sfx])
    np.testing.assert_array_equal(sc, g[tag + "_sc" + sfx])
    np.testing.assert_array_equal(inter["peaks"][:, 0, 0].astype(np.int32), g[tag + "_peaks_n"])


def test_post_v2_known_answers():
    """Torso box and 3-D lift restatements (skeleton.py:89-113, 130-160): hand-computed cases."""
    from oracle import post_v2
    torso = [1, 2, 5, 8, 9, 12]
    kp = np.zeros((3, 25, 3), np.float32)
    kp[0, 1] = (10, 20, .9); kp[0, 8] = (14, 50, .8); kp[0, 2] = (5, 25, .7); kp[0, 0] = (100, 100, .9)   # nose ignored
    kp[1, 5] = (7.5, 9.25, .5)                                    # single joint: degenerate box -> +1 on both sides
    bb = post_v2.torso_bbox(kp, torso)                            # person 2: no torso joint -> [0,0,1,1]
    np.testing.assert_array_equal(bb, np.array([[5, 20, 14, 50], [7.5, 9.25, 8.5, 10.25], [0, 0, 1, 1]], np.float32))
    assert bb.dtype == np.float32
    depth = np.arange(8 * 10, dtype=np.uint16).reshape(8, 10) * 10
    sk = np.array([[4.7, 3.2, .9], [0.4, 0.9, .8], [50, 50, .5]], np.float32)
    out = post_v2.lift_3d(sk, depth, fx=2.0, fy=4.0, cx=1.0, cy=2.0, depth_scale=1e-3, patch_offset=2)
    # joint 0: rows int(1.2)=1..int(5.2)=5 -> 1..4, cols int(2.7)=2..int(6.7)=6 -> 2..5: mean of (10*(10*r+c))
    avg0 = np.mean([10 * (10 * r + c) for r in range(1, 5) for c in range(2, 6)])
    x, y = float(np.float32(4.7)), float(np.float32(3.2))
    np.testing.assert_array_equal(out[0], [(x - 1.0) / 2.0 * avg0 * 1e-3, (y - 2.0) / 4.0 * avg0 * 1e-3, avg0 * 1e-3])
    # joint 1: int(-1.6) = -1 -> 0 (truncation, not floor); rows 0..int(2.9)=2 -> 0,1; cols 0..int(2.4)=2 -> 0,1
    avg1 = np.mean([0, 10, 100, 110])
    assert out[1, 2] == avg1 * 1e-3
    assert np.isnan(out[2]).all()                                 # patch outside the image: nanmean of nothing


def test_feature_oracle_pinned_by_the_reference_and_known_answers():
    """oracle/features.py: resize_tensor equals the REFERENCE's own outputs (tests/golden/callers_ref.npz, made by
    openpose/native/python/utils.py); heatmaps_scale known answers; crop rules of create_detections."""
    import os
    from oracle import features
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "callers_ref.npz"))
    for i in range(5):
        x, y = (int(v) for v in g[f"rt{i}_xy"])
        assert np.array_equal(features.resize_tensor(g[f"rt{i}_in"], x, y), g[f"rt{i}_out"])
    v = np.array([-1.5, -1.0, -0.25, 0.0, 0.25, 0.999, 1.0, 1.7], np.float32)
    assert features.scale_values(v, True, 3) is not None and np.array_equal(features.scale_values(v, True, 3), v)
    assert features.scale_values(v, True, 1).tolist() == [0, 0, 0, 0, 0.25, np.float32(0.999), 1, 1]
    assert features.scale_values(v, True, 0).tolist() == [-1, -1, -1, -1, -0.5, np.float32(0.999) * 2 - 1, 1, 1]
    assert features.scale_values(v, True, 2).tolist() == [0, 0, 0, 0, 64, 255, 255, 255]          # int(63.75 + .5)
    assert features.scale_values(v, False, 1).tolist() == [0, 0, 0.375, 0.5, 0.625, np.float32(0.999) * np.float32(0.5) + np.float32(0.5), 1, 1]
    assert features.scale_values(v, False, 2).tolist() == [0, 0, 96, 129, 161, 255, 255, 255]
    assert features.scale_values(v, False, 0).tolist() == [-1, -1, -0.25, 0, 0.25, np.float32(0.999), 1, 1]
    # crop + resample: a ramp map makes the sampled source indices visible
    H, W = 40, 60
    yy, xx = np.mgrid[0:H, 0:W]
    hm = np.stack([yy, xx], axis=-1).astype(np.float32)
    box = np.array([[10.0, 4.0, 31.5, 35.2]], np.float32)               # l, t, r, b in image pixels
    f = features.detection_features(box, hm, (120, 80), (25, 50))[0].reshape(50, 25, 2)   # s_w = s_h = 0.5
    assert not f[0].any() and not f[:, 0].any()
    # crop rows 2:18 (h 16), columns 5:16 (w 11): out[i+1][j+1] = crop[1 + int(i/(50/15)), 1 + int(j/(25/10))]
    assert f[1, 1].tolist() == [3.0, 6.0] and f[49, 24].tolist() == [2 + 1 + int(48 / (50 / 15)), 5 + 1 + int(23 / 2.5)]
    wide = features.detection_features(np.array([[10.0, 4.0, 10.5, 35.0]], np.float32), hm, (120, 80))[0]
    assert isinstance(wide, IndexError)


# ---- the CUDA build's connector: pafPtrIntoVector + pafVectorIntoPeopleVector ----------------------
def _peaks_with(entries):
    """{part: [(x, y, score), ...]} -> upstream peaks array [25, 128, 3]."""
    P1 = BODY_25.max_peaks + 1
    peaks = np.zeros((25, P1, 3), np.float32)
    for part, lst in entries.items():
        peaks[part, 0, 0] = len(lst)
        for i, v in enumerate(lst):
            peaks[part, i + 1] = v
    return peaks


def _idx(part, i):
    return part * 3 * (BODY_25.max_peaks + 1) + i * 3 + 2


def _limb(a, b):
    return BODY_25.pairs.index((a, b))


def test_cuda_connector_sorted_list_and_greedy_conflict():
    # the scene of test_sorted_greedy_and_assembly_rules: two necks, two mid-hips, PAF straight down
    peaks = _peaks_with({1: [(100, 100, .9), (300, 100, .8)], 8: [(100, 200, .7), (200, 200, .6)]})
    maps = np.zeros((78, 368, 656), np.float32)
    maps[BODY_25.n_heat + BODY_25.mapidx[0][1]] = 1.0
    conns = decode_v2.paf_ptr_into_vector(peaks, maps, BODY_25)
    # all four pairs are accepted; order = descending paf + 0.1 * (scoreA + scoreB)
    assert [(c[2], c[3], c[4]) for c in conns] == [(0, 1, 1), (0, 1, 2), (0, 2, 2), (0, 2, 1)]
    f = np.float32
    assert conns[0][0] == float(f(f(f(1.0) + f(f(0.1) * f(.9))) + f(f(0.1) * f(.7)))) and conns[0][1] == 1.0
    rows, scores = decode_v2.paf_vector_into_people_vector(conns, peaks, BODY_25)
    # neck0-hip0 founds person 0; neck0-hip1 is dropped (person 0 has a hip); neck1-hip1 founds person 1;
    # neck1-hip0 joins two people that both have a neck and a hip -> no merge
    assert rows[:, [1, 8, 25]].tolist() == [[_idx(1, 1), _idx(8, 1), 2], [_idx(1, 2), _idx(8, 2), 2]]
    assert scores[0] == f(f(f(.9) + f(.7)) + f(1.0))
    assert scores[1] == f(f(f(.8) + f(.6)) + f(conns[2][1]))


def test_cuda_connector_extend_merge_and_cycle_rules():
    f = np.float32
    peaks = _peaks_with({p: [(10 * p + 5, 50, 0.5 + 0.01 * p), (10 * p + 5, 150, 0.4 + 0.01 * p)]
                         for p in (0, 1, 2, 8, 9, 10, 15, 17)})
    s = lambda part, i: peaks[part, i, 2]                                        # noqa: E731
    # hand-made, already sorted list (total, paf, limb, iA, iB)
    conns = [
        (0.99, 0.90, _limb(1, 8), 1, 1),      # 1. neither assigned -> person 0 = {neck1, hip1}
        (0.98, 0.89, _limb(9, 10), 1, 1),     # 1. person 1 = {rhip1, rknee1}
        (0.97, 0.88, _limb(1, 8), 1, 2),      # 2. neck1 belongs to person 0, which has a mid-hip -> ignored
        (0.96, 0.87, _limb(8, 9), 1, 1),      # 5. hip1 in person 0, rhip1 in person 1, disjoint -> merge into 0
        (0.95, 0.86, _limb(1, 2), 1, 1),      # 2. extend person 0 with rshoulder1
        (0.94, 0.85, _limb(1, 0), 1, 1),      #    ... nose1
        (0.93, 0.84, _limb(0, 15), 1, 1),     #    ... reye1
        (0.92, 0.83, _limb(15, 17), 1, 1),    #    ... rear1
        (0.91, 0.82, _limb(2, 17), 1, 1),     # 4. rshoulder1 and rear1 both in person 0 -> score += paf only
        (0.90, 0.81, _limb(1, 8), 2, 2),      # 1. person 2 = {neck2, hip2}
        (0.89, 0.80, _limb(9, 10), 2, 2),     # 1. person 3 = {rhip2, rknee2}
        (0.88, 0.79, _limb(8, 9), 2, 1),      # 5. hip2 (person 2) - rhip1 (person 0): both have a mid-hip -> no merge
        (0.87, 0.78, _limb(8, 9), 2, 2),      # 5. person 2 + person 3 disjoint -> merge into 2
        (0.86, 0.77, _limb(1, 2), 2, 1),      # 3.-like: rshoulder1 is person 0's, neck2 is person 2's: both have
                                              #    a neck -> no merge
    ]
    rows, scores = decode_v2.paf_vector_into_people_vector(conns, peaks, BODY_25)
    assert rows.shape == (2, 26)                       # persons 1 and 3 were merged away and removed
    want0 = {1: 1, 8: 1, 9: 1, 10: 1, 2: 1, 0: 1, 15: 1, 17: 1}
    for part in range(25):
        assert rows[0, part] == (_idx(part, want0[part]) if part in want0 else 0)
    assert rows[0, 25] == 8
    sc = f(f(s(1, 1) + s(8, 1)) + f(.90))                                     # create
    s1 = f(f(s(9, 1) + s(10, 1)) + f(.89))
    sc = f(sc + f(s1 + f(.87)))                                               # merge: s0 += s1 + paf
    for part, paf in ((2, .86), (0, .85), (15, .84), (17, .83)):
        sc = f(sc + f(s(part, 1) + f(paf)))                                   # extend: += peak + paf
    sc = f(sc + f(.82))                                                       # cycle: += paf
    assert scores[0] == sc
    assert sorted(np.nonzero(rows[1, :25])[0].tolist()) == [1, 8, 9, 10] and rows[1, 25] == 4
    assert rows[1, 8] == _idx(8, 2) and rows[1, 9] == _idx(9, 2)
    s2 = f(f(s(1, 2) + s(8, 2)) + f(.81))
    s3 = f(f(s(9, 2) + s(10, 2)) + f(.80))
    assert scores[1] == f(s2 + f(s3 + f(.78)))
    # a B end that is assigned while A is free joins B's person (case 3)
    rows, _ = decode_v2.paf_vector_into_people_vector(
        [(0.9, 0.8, _limb(8, 9), 1, 1), (0.8, 0.7, _limb(1, 8), 2, 1)], peaks, BODY_25)
    assert rows.shape[0] == 1 and rows[0, 1] == _idx(1, 2) and rows[0, 25] == 3
    # min-parts filter on the merged result: 8 parts, score/8 >= 0.4 -> kept; the 4-part person too
    rows, scores = decode_v2.paf_vector_into_people_vector(conns, peaks, BODY_25)
    assert decode_v2.filter_people(rows, scores, BODY_25) == [0, 1]


def test_cuda_connector_end_to_end_differs_from_cpu_connector_only_in_assembly():
    from posehip import synth
    scene = synth.synth_net_output(101, 2, BODY_25, 46, 82)
    k0, s0, i0 = decode_v2.decode(scene, BODY_25, 368, 656, 1.0, True, connector=0)
    k1, s1, i1 = decode_v2.decode(scene, BODY_25, 368, 656, 1.0, True, connector=1)
    np.testing.assert_array_equal(i0["peaks"], i1["peaks"])
    assert len(s0) == len(s1) == 2
    # every limb the per-limb greedy keeps is also in the global list (with the same PAF score)
    have = {(c[2], c[3], c[4]): c[1] for c in i1["connections"]}
    off = 3 * (BODY_25.max_peaks + 1)
    for k, lst in enumerate(i0["connections"]):
        for ia, ib, sc in lst:
            assert have[(k, (ia % off) // 3, (ib % off) // 3)] == float(sc)
    # the close-keypoint threshold is float in the CUDA build, double in the CPU build
    assert decode_v2._close_thr(656, 368, True) == float(np.float32(np.sqrt(np.float32(656 * 368))) / np.float32(150))
    assert decode_v2._close_thr(656, 368, False) == np.sqrt(656.0 * 368.0) / 150.0
