"""CPU: the V1 oracle (transcription of openpose/opencv/utils.py + restated cv2
primitives) against known-answer cases and the golden scenes."""
import json
import os

import numpy as np
import pytest

from oracle import decode_v1
from posehip import tables

G = os.path.join(os.path.dirname(__file__), "golden")
BP, PP = tables.get_body_parts_and_pose_pairs("COCO")


def test_tables_equal_reference_common_py():
    ref = json.load(open(os.path.join(G, "tables_ref.json")))
    assert tables.COLORS == ref["COLORS"] and tables.MAPIDX == ref["MAPIDX"]
    for ds in ("COCO", "MPI"):
        bp, pp = tables.get_body_parts_and_pose_pairs(ds)
        assert bp == ref[ds]["body_parts"] and pp == ref[ds]["pose_pairs"]
    assert len(tables.MAPIDX) == 16 and len(PP) == 17          # LEye->LEar is never decoded (Appendix B.1)
    with pytest.raises(Exception):
        tables.get_body_parts_and_pose_pairs("BODY_25")


def test_resize_linear_kats():
    ramp = np.tile(np.arange(4, dtype=np.float32), (4, 1))
    up = decode_v1.resize_linear(ramp, 8, 8)
    np.testing.assert_allclose(up[0], [0, 0.25, 0.75, 1.25, 1.75, 2.25, 2.75, 3], atol=1e-6)   # half-pixel centres
    assert (up == up[0]).all()
    same = decode_v1.resize_linear(ramp, 4, 4)
    np.testing.assert_array_equal(same, ramp)                                               # identity
    rng = np.random.default_rng(1)
    m = rng.normal(size=(46, 82)).astype(np.float32)
    full = decode_v1.resize_linear(m, 1920, 1080)                                            # anisotropic 23.4x/23.5x
    xs, ys = np.array([0, 5, 1919, 960, 1234]), np.array([0, 1079, 1079, 540, 77])
    np.testing.assert_array_equal(decode_v1.resize_linear_at(m, 1920, 1080, xs, ys), full[ys, xs])


def test_gaussian3_reflect101():
    m = np.zeros((5, 6), np.float32)
    m[0, 0] = 16
    g = decode_v1.gaussian3(m)
    np.testing.assert_array_equal(g[:2, :2], [[4, 2], [2, 1]])   # corner impulse: 101-reflection does not repeat the edge
    assert g.sum() == 9 and g[2:].sum() == 0
    m = np.zeros((5, 6), np.float32)
    m[1, 1] = 16
    np.testing.assert_array_equal(decode_v1.gaussian3(m)[:3, :3], [[4, 4, 2], [4, 4, 2], [2, 2, 1]])  # tap at -1 -> +1
    m = np.zeros((5, 6), np.float32)
    m[2, 3] = 16
    np.testing.assert_array_equal(decode_v1.gaussian3(m)[1:4, 2:5], [[1, 2, 1], [2, 4, 2], [1, 2, 1]])


def test_blob_peaks_order_ties_and_score_source():
    m = np.zeros((40, 60), np.float32)
    m[5, 5] = 1.0
    m[30, 40] = 0.8
    m[30, 42] = 0.8            # merges with its twin after the blur; first raster max wins
    kps = decode_v1.blob_peaks(m, 0.1)
    assert [(k[0], k[1]) for k in kps] == [(40, 30), (5, 5)]    # findContours order: last blob first;
    assert kps[0][2] == np.float32(0.8) and kps[1][2] == 1.0    # smoothed tie (40 vs 41): first raster wins
    ring = np.zeros((20, 20), np.float32)
    ring[9, 10] = ring[11, 10] = ring[10, 9] = ring[10, 11] = 0.5
    (x, y, s), = decode_v1.blob_peaks(ring, 0.1)
    assert (x, y) == (10, 10) and s == 0.0   # score is read from the UN-smoothed map at the smoothed arg-max (utils.py:22)
    assert decode_v1.blob_peaks(np.full((8, 8), 0.1, np.float32), 0.1) == []   # strict >


def test_rounding_never_hits_a_tie():
    """Samples are a + k*(b-a)/9 with integer a,b: never on a half-integer, so
    banker's rounding (utils.py:91-94) == any round-to-nearest."""
    d = np.arange(-2048, 2049)[:, None].astype(np.float64)
    k = np.arange(10)[None, :].astype(np.float64)
    x = k * (d / 9.0)
    frac = np.abs(x - np.floor(x) - 0.5)
    assert frac.min() > 1.0 / 19


def _pred_with_limb(ax, ay, bx, by, hs=46, ws=46):
    pred = np.zeros((1, 57, hs, ws), np.float32)
    d = np.array([bx - ax, by - ay], np.float64)
    d /= np.linalg.norm(d)
    pred[0, tables.MAPIDX[0][0]] = d[0]
    pred[0, tables.MAPIDX[0][1]] = d[1]
    return pred


def test_valid_pairs_rules():
    pred = _pred_with_limb(10, 10, 30, 10)
    det = [[] for _ in range(18)]
    det[1] = [(10, 10, 0.9, 0), (10, 40, 0.8, 1)]               # two necks
    det[2] = [(30, 10, 0.7, 2), (10, 10, 0.6, 3)]               # RShoulder candidates; the 2nd coincides with neck 0
    vp, inv = decode_v1.get_valid_pairs(pred, det, PP, BP, 46, 46, tables.MAPIDX)
    assert inv == list(range(1, 16))                            # limbs with an empty side are "invalid"
    rows = vp[0]
    # neck 0 -> shoulder 2 with score 1 (zero-length candidate skipped); neck 1 also claims a shoulder:
    # no uniqueness constraint on B (Appendix B.4)
    assert rows.shape[1] == 3 and rows[0, 0] == 0 and rows[0, 1] == 2 and abs(rows[0, 2] - 1.0) < 1e-6
    assert all(r[1] in (2, 3) for r in rows)
    # both sides present but PAF opposes -> empty [0,3] array, NOT invalid (Appendix B.5)
    vp, inv = decode_v1.get_valid_pairs(-pred, det, PP, BP, 46, 46, tables.MAPIDX)
    assert 0 not in inv and vp[0].shape == (0, 3)


def test_personwise_rules():
    kl = np.array([[0, 0, .9], [0, 0, .8], [0, 0, .7], [0, 0, .6]])
    vp = [np.zeros((0, 3)) for _ in range(16)]
    vp[0] = np.array([[0., 2., 0.5]])                           # Neck0 - RShoulder2
    vp[2] = np.array([[2., 3., 0.25]])                          # RShoulder2 - RElbow3: extends person 0
    vp[1] = np.array([[1., 3., 0.1]])                           # Neck1 - LShoulder(3): new person
    people = decode_v1.get_personwise_keypoints(vp, [], kl, BP, PP, tables.MAPIDX)
    assert len(people) == 2
    assert people[0][1] == 0 and people[0][2] == 2 and people[0][3] == 3
    assert people[0][-1] == pytest.approx((.9 + .7 + .5) + (.6 + .25))
    assert people[1][1] == 1 and people[1][5] == 3 and (people[1] == -1).sum() == 16


def test_postprocess_single():
    pred = np.zeros((1, 19, 46, 46), np.float32)
    pred[0, 0, 10, 20] = 0.9
    pred[0, 1, 5, 5] = 0.05
    pts = decode_v1.postprocess_single(pred, 19, 480, 368, 0.1)
    assert pts[0] == (int(480 * 20 / 46), int(368 * 10 / 46)) and pts[1] is None


@pytest.mark.parametrize("tag", ["p3_480", "p4_656", "p2_1080"])
def test_golden_scenes(tag):
    g = np.load(os.path.join(G, "v1_scenes.npz"))
    pred = g[tag + "_net"].astype(np.float32)[None]
    W, H = (int(v) for v in g[tag + "_size"])
    kl, people = decode_v1.postprocess_multi(pred, BP, PP, tables.MAPIDX, W, H, 0.1)
    np.testing.assert_array_equal(kl, g[tag + "_kl"])
    np.testing.assert_array_equal(np.stack(people) if people else np.zeros((0, 19)), g[tag + "_people"])


def test_preprocess_restatement_known_answers_and_fixture():
    """K1 oracle (cv2.resize INTER_LINEAR u8, 11-bit fixed point): hand-computed cases + the G6 fixture."""
    import os
    from oracle import preprocess
    img = np.array([[[0, 0, 0], [255, 255, 255]]], np.uint8)[None]           # 1x2 -> 1x4
    np.testing.assert_array_equal(preprocess.cv_resize_u8(img, 4, 1)[0, 0, :, 0], [0, 64, 191, 255])
    same = np.arange(2 * 5 * 7 * 3, dtype=np.uint8).reshape(2, 5, 7, 3)
    np.testing.assert_array_equal(preprocess.cv_resize_u8(same, 7, 5), same)  # same size: identity
    flat = np.full((1, 9, 11, 3), 137, np.uint8)
    assert (preprocess.cv_resize_u8(flat, 40, 24) == 137).all()               # weights sum to 2048: constants survive
    blob = preprocess.blob_from_image(flat, 8, 8)
    assert blob.shape == (1, 3, 8, 8) and blob.dtype == np.float32
    np.testing.assert_array_equal(blob, np.float32(137) * np.float32(1 / 255.0))
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "real_images.npz"))
    for tag in ("pillar_text", "coco_192"):
        got = preprocess.cv_resize_u8(g[tag + "_bgr"][None], 656, 368)[0]
        np.testing.assert_array_equal(got, g[tag + "_resized_656x368"])


# ---- native-flavour preprocess (OpenPose cvMatToOpInput = cv::warpAffine in fixed point) -----------------------
def test_native_preprocess_oracle_known_answers():
    from oracle import preprocess as P
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, (1, 9, 13, 3), dtype=np.uint8)
    # scale 1 into a larger canvas: the frame is copied to the top-left corner, the rest is the border value 0
    out = P.op_resize_fixed_aspect(np.pad(img, ((0, 0), (0, 0), (0, 0), (0, 0))), 13, 9)
    np.testing.assert_array_equal(out, img)
    wide = P.warp_affine_scale_u8(img, 1.0, 16, 9)
    np.testing.assert_array_equal(wide[:, :, :13], img)
    assert (wide[:, :, 13:] == 0).all()
    # exact 2:1 down-scale: s = (Wn-1)/(W0-1) is NOT 1/2 (OpenPose's "-1" convention), so pixel x maps to
    # x * (W0-1)/(Wn-1); with a 17 -> 9 pixel row that is exactly 2x and every output pixel is a source pixel
    row = rng.integers(0, 256, (1, 17, 17, 3), dtype=np.uint8)
    half = P.op_resize_fixed_aspect(row, 9, 9)
    np.testing.assert_array_equal(half, row[:, ::2, ::2])
    # half-way phases: 5 -> 9 pixels is s = 2 > 1 -> INTER_CUBIC; odd outputs sit half-way between sources:
    # weights (-0.09375, 0.59375, 0.59375, -0.09375) = OpenCV's cubic (A = -0.75) at 0.5, taps outside are 0
    ramp = np.zeros((1, 5, 5, 3), np.uint8)
    ramp[0, :, :, :] = (np.arange(5) * 40)[None, :, None]
    up = P.op_resize_fixed_aspect(ramp, 9, 9)
    np.testing.assert_array_equal(up[0, 0, ::2, 0], np.arange(5) * 40)
    assert up[0, 0, 3, 0] == 60            # interior: the cubic reproduces a linear ramp
    assert up[0, 0, 1, 0] == int(np.floor(0.59375 * 40 - 0.09375 * 80 + 0.5))     # left tap outside -> 0
    # the fixed-point tables: every 2-D weight set sums to 1.0 (32768); entry (0, 0) is the saturated 32767 + 1
    lin, cub = P._inter_tab_2d(False), P._inter_tab_2d(True)
    assert (lin.reshape(1024, -1).sum(1) == 32768).all() and (cub.reshape(1024, -1).sum(1) == 32768).all()
    assert lin[0].tolist() == [[32767, 0], [0, 1]] and lin[33].tolist() == [[30752, 992], [992, 32]]
    assert cub[0, 1, 1] == 32767 and cub[0, 2, 2] == 1
    # normalisation: (x / 256) - 0.5, CHW
    blob = P.cv_mat_to_op_input(img, 13, 9)
    assert blob.shape == (1, 3, 9, 13)
    np.testing.assert_array_equal(blob[0, :, 2, 3], img[0, 2, 3].astype(np.float32) / 256 - 0.5)
    assert P.op_net_input_size("-1x368", 1920, 1080) == (656, 368)
    assert P.op_scale_factor(1920, 1080, 656, 368) == min(655 / 1919, 367 / 1079)


# ---- getKeypoints literally: findContours(RETR_TREE, CHAIN_APPROX_SIMPLE) + fillConvexPoly -------------
def test_find_contours_tree_known_answers():
    from oracle import decode_v1 as D
    m = np.zeros((8, 10), np.uint8)
    m[1:4, 1:4] = 1                     # a square, found first
    m[1:6, 5:10] = 1
    m[2:5, 6:9] = 0                     # a ring, found second (same first row, further right)
    cs = [c.tolist() for c in D.find_contours_tree(m)]
    # list order: the border found LAST comes first; a hole border follows its outer border; vertices start at the
    # border's first raster pixel and run counter-clockwise on screen (outer) / clockwise (hole); a hole border
    # consists of FOREGROUND pixels, so its corners are cut diagonally
    assert cs == [[[5, 1], [5, 5], [9, 5], [9, 1]],
                  [[5, 2], [6, 1], [8, 1], [9, 2], [9, 4], [8, 5], [6, 5], [5, 4]],
                  [[1, 1], [1, 3], [3, 3], [3, 1]]]
    # a blob inside a hole: outer, its hole, the inner blob (single pixel = single vertex)
    n = np.zeros((9, 9), np.uint8)
    n[1:8, 1:8] = 1
    n[2:7, 2:7] = 0
    n[4, 4] = 1
    cs = [c.tolist() for c in D.find_contours_tree(n)]
    assert cs == [[[1, 1], [1, 7], [7, 7], [7, 1]],
                  [[1, 2], [2, 1], [6, 1], [7, 2], [7, 6], [6, 7], [2, 7], [1, 6]],
                  [[4, 4]]]
    # diamond: four vertices, starts at the top
    yy, xx = np.mgrid[0:9, 0:9]
    d = (abs(yy - 4) + abs(xx - 4) <= 3).astype(np.uint8)
    assert [c.tolist() for c in D.find_contours_tree(d)] == [[[4, 1], [1, 4], [4, 7], [7, 4]]]
    # pixels touching the image border are handled (zero frame added, as cv2 does since 3.2)
    e = np.ones((3, 4), np.uint8)
    assert [c.tolist() for c in D.find_contours_tree(e)] == [[[0, 0], [0, 2], [3, 2], [3, 0]]]
    assert D.find_contours_tree(np.zeros((4, 4), np.uint8)) == []


def test_fill_convex_poly_known_answers():
    from oracle import decode_v1 as D
    yy, xx = np.mgrid[0:9, 0:9]
    d = (abs(yy - 4) + abs(xx - 4) <= 3)
    np.testing.assert_array_equal(D.fill_convex_poly(d.shape, D.find_contours_tree(d)[0]), d)     # convex: exact
    sq = np.zeros((6, 7), bool)
    sq[1:5, 2:6] = True
    np.testing.assert_array_equal(D.fill_convex_poly(sq.shape, D.find_contours_tree(sq)[0]), sq)
    # a U with 5-pixel arms: the two edge walkers follow the left arm only; of the right arm just the outline is
    # painted, its interior is missed (the polygon is not convex)
    u = np.zeros((18, 21), bool)
    u[1:15, 1:6] = True
    u[1:15, 13:18] = True
    u[11:15, 1:18] = True
    f = D.fill_convex_poly(u.shape, D.find_contours_tree(u)[0])
    assert not (f & ~u).any()
    missed = np.argwhere(u & ~f)
    assert len(missed) == 3 * 9 and missed[:, 1].min() == 14 and missed[:, 1].max() == 16     # interior of the right arm
    assert missed[:, 0].min() == 2 and missed[:, 0].max() == 10


def test_get_keypoints_literal_vs_connected_components():
    """Where the literal getKeypoints (contours + fillConvexPoly) and the connected-component form the GPU kernels
    implement agree, and the three documented cases where they do not."""
    from oracle import decode_v1 as D
    f32 = np.float32
    yy, xx = np.mgrid[0:24, 0:40].astype(f32)

    def bump(x0, y0, a, s=3.0):
        return (a * np.exp(-((xx - x0) ** 2 + (yy - y0) ** 2) / (2 * s * s))).astype(f32)
    # isolated bumps and two MERGED people (one peanut-shaped blob): identical, one keypoint per blob
    for m in (bump(10, 10, 0.8), np.maximum(bump(8, 12, 0.8), bump(30, 9, 0.7)),
              np.maximum(bump(12, 12, 0.8), bump(24, 10, 0.9))):
        assert D.blob_peaks(m, 0.1) == D.blob_peaks_cc(m, 0.1)
    merged = D.blob_peaks(np.maximum(bump(12, 12, 0.8), bump(24, 10, 0.9)), 0.1)
    assert [(x, y) for x, y, _ in merged] == [(24, 10)]                  # the lower person of the pair is lost (V1)
    # ring blob: RETR_TREE also returns the hole border -> an extra keypoint on the ring's inner edge whose score
    # is read from the UNSMOOTHED map (0 here); every later keypoint id shifts by one
    r = np.zeros((16, 16), f32)
    r[3:13, 3:13] = 0.5
    r[6:10, 6:10] = 0
    r[4, 11] = 0.9
    assert D.blob_peaks(r, 0.1) == [(11, 4, f32(0.9)), (7, 6, f32(0.0))]
    assert D.blob_peaks_cc(r, 0.1) == [(11, 4, f32(0.9))]
    # U-shaped blob whose maximum sits inside the arm fillConvexPoly misses: the literal form reports the first
    # raster maximum of what was painted instead
    u = np.zeros((20, 22), f32)
    u[2:16, 2:7] = 0.5
    u[2:16, 14:19] = 0.5
    u[12:16, 2:19] = 0.5
    u[5, 16] = 0.9
    assert D.blob_peaks(u, 0.1) == [(3, 3, f32(0.5))]
    assert D.blob_peaks_cc(u, 0.1) == [(16, 5, f32(0.9))]


@pytest.mark.parametrize("tag", ["p3_480", "p4_656", "p2_1080"])
def test_golden_scenes_literal_blobs_equal_connected_components(tag):
    """On the committed V1 scenes (which the GPU kernels reproduce bit for bit) the literal getKeypoints gives
    the same keypoints in the same order as the connected-component form: GPU == literal there, transitively."""
    import os
    from oracle import decode_v1 as D
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "v1_scenes.npz"))
    W, H = (int(v) for v in g[tag + "_size"])
    net = g[tag + "_net"].astype(np.float32)
    for part in range(18):
        heat = D.resize_linear(net[part], W, H)
        assert D.blob_peaks(heat, 0.1) == D.blob_peaks_cc(heat, 0.1)
