"""GPU parity: V1 decode kernels (resize+blur+runs, blobs, PAF pairs, assembly)
vs the transcription of openpose/opencv/utils.py in oracle/decode_v1.py, through
the C ABI.  Keypoint coordinates / ids / person rows are bit-exact; scores too
(same IEEE operations in the same order)."""
import numpy as np
import pytest

from posehip import synth
from posehip.tables import COCO_18, MAPIDX, get_body_parts_and_pose_pairs

pytestmark = pytest.mark.gpu

BODY_PARTS, POSE_PAIRS = get_body_parts_and_pose_pairs("COCO")


def _oracle(pred, W, H, thr=0.1):
    from oracle import decode_v1
    # blobs="opencv": the literal findContours(RETR_TREE) + fillConvexPoly walk of
    # getKeypoints (utils.py:14-22), holes and under-filled concave blobs included
    return decode_v1.postprocess_multi(pred, BODY_PARTS, POSE_PAIRS, MAPIDX, W, H, thr,
                                       return_intermediates=True, blobs="opencv")


def _check(got, ref):
    klist, people, ex = got
    rk, rp, ri = ref
    assert klist.shape == rk.shape
    np.testing.assert_array_equal(klist, rk)
    assert ex["invalid_pairs"] == ri["invalid_pairs"]
    for k, (a, b) in enumerate(zip(ex["valid_pairs"], ri["valid_pairs"])):
        if k in ri["invalid_pairs"]:
            continue
        np.testing.assert_array_equal(np.asarray(a).reshape(-1, 3), np.asarray(b).reshape(-1, 3))
    assert len(people) == len(rp)
    for a, b in zip(people, rp):
        np.testing.assert_array_equal(a, b)


@pytest.mark.parametrize("hs,ws,W,H,people", [
    (46, 46, 368, 368, 2),      # net-resolution decode
    (46, 46, 480, 480, 3),      # the reference bench: 480x480 image, 368x368 net (opencv/inference.py:67)
    (46, 82, 656, 368, 5),
    (46, 82, 641, 359, 2),      # non-integer, anisotropic scale
    (46, 82, 1920, 1080, 3),    # BASELINE config 3: MOT17 1080p
    (46, 82, 1920, 1080, 20),   # ... crowded, at the reference's default conf_thres = 0.1 (265 keypoints, 34 rows)
    (46, 82, 300, 200, 0),      # empty scene: every limb "No Connection"
])
def test_decode_v1_matches_oracle(engine, hs, ws, W, H, people):
    pred = synth.synth_net_output(40 + people, people, COCO_18, hs, ws)[None]
    engine.set_net_out(pred)
    got = engine.decode_v1(BODY_PARTS, POSE_PAIRS, MAPIDX, 1, W, H, 0.1)
    if people == 20:
        wave, group = engine.decode_v1_walks()
        assert wave > 0 and group == 0
    ref = _oracle(pred, W, H)
    _check(got[0], ref)
    if people:
        assert len(ref[1]) > 0 and ref[0].shape[0] >= 10


def test_decode_v1_batch_and_crowd(engine):
    preds = np.stack([synth.synth_net_output(70 + i, p, COCO_18, 46, 82) for i, p in enumerate([12, 1, 0, 6])])
    engine.set_net_out(preds)
    got = engine.decode_v1(BODY_PARTS, POSE_PAIRS, MAPIDX, 4, 656, 368, 0.1)
    for b in range(4):
        _check(got[b], _oracle(preds[b:b + 1], 656, 368))


def test_blob_edge_cases(engine):
    """Blobs touching the image border, a blob spanning a 256-px tile seam, two
    maxima of equal height (first raster wins), a plateau at the threshold."""
    pred = np.zeros((1, 57, 46, 82), np.float32)
    pred[0, 0, 0, 0] = 0.9            # corner
    pred[0, 0, 45, 81] = 0.8          # opposite corner
    pred[0, 1, 20, 30:36] = 0.7       # long flat ridge crossing x = 256 at 8x
    pred[0, 2, 10, 10] = 0.6
    pred[0, 2, 10, 14] = 0.6          # twin peaks merging into one blob after blur
    pred[0, 3, 30:34, 40:44] = 0.1    # plateau at the threshold: rounding decides pixel by pixel
    engine.set_net_out(pred)
    got = engine.decode_v1(BODY_PARTS, POSE_PAIRS, MAPIDX, 1, 656, 368, 0.1)
    ref = _oracle(pred, 656, 368)
    _check(got[0], ref)
    assert ref[0].shape[0] >= 4


@pytest.fixture(params=["wave", "group"])
def blob_walk(request, monkeypatch):
    """Both walks of the blobs that need the literal contour + fill: a whole wave per blob (default) and the
    one-group border following it falls back to (POSEHIP_V1_FAST=0, read at every call)."""
    if request.param == "group":
        monkeypatch.setenv("POSEHIP_V1_FAST", "0")
    else:
        monkeypatch.delenv("POSEHIP_V1_FAST", raising=False)
    return request.param


def _check_walk(engine, blob_walk, min_blobs=1):
    wave, group = engine.decode_v1_walks()
    assert wave + group >= min_blobs
    if blob_walk == "group":
        assert wave == 0
    return wave, group


def _shape_scene():
    """Heat maps whose blobs are NOT one run per row: rings (hole contours are
    keypoints too, utils.py:17), a ring with an island in its hole (RETR_TREE
    nesting order), U / C / S / X shapes that fillConvexPoly under-fills, two
    holes in one blob, a ring on the image border."""
    pred = np.zeros((1, 57, 46, 82), np.float32)
    pred[0, 0, 5:10, 5:10] = 0.9; pred[0, 0, 6:9, 6:9] = 0.0
    pred[0, 1, 10:21, 10:21] = 0.8; pred[0, 1, 12:19, 12:19] = 0.0; pred[0, 1, 15, 15] = 0.95
    pred[0, 2, 20:30, 40] = 0.5; pred[0, 2, 29, 40:50] = 0.5; pred[0, 2, 20:30, 49] = 0.6; pred[0, 2, 21, 49] = 0.9
    pred[0, 3, 5, 30:38] = 0.7; pred[0, 3, 5:15, 37] = 0.7; pred[0, 3, 14, 30:38] = 0.75
    pred[0, 4, 30, 10:20] = 0.6; pred[0, 4, 30:36, 10] = 0.6; pred[0, 4, 35, 10:20] = 0.6
    pred[0, 4, 35:41, 19] = 0.6; pred[0, 4, 40, 10:20] = 0.65
    pred[0, 5, 5:12, 50:63] = 0.8; pred[0, 5, 7:10, 52:55] = 0.0; pred[0, 5, 7:10, 58:61] = 0.0; pred[0, 5, 8, 70] = 0.9
    for i in range(8):
        pred[0, 6, 20 + i, 60 + i] = 0.7
    pred[0, 7, 0:6, 0:6] = 0.8; pred[0, 7, 2:4, 2:4] = 0.0
    for i in range(10):
        pred[0, 8, 30 + i, 60 + i] = 0.7
        pred[0, 8, 39 - i, 60 + i] = 0.7
    return pred


@pytest.mark.parametrize("W,H", [(656, 368), (480, 480), (1280, 720)])
def test_literal_contours_shapes(engine, W, H, blob_walk):
    from oracle import decode_v1
    pred = _shape_scene()
    engine.set_net_out(pred)
    got = engine.decode_v1(BODY_PARTS, POSE_PAIRS, MAPIDX, 1, W, H, 0.1)
    wave, group = _check_walk(engine, blob_walk, 5)
    if blob_walk == "wave":
        assert group == 0          # rings, nested rings, U / C / S / X shapes: none of them trips a guard
    ref = _oracle(pred, W, H)
    _check(got[0], ref)
    # the scene really exercises the literal path: the connected-component
    # arg-max rule gives a different keypoint list
    cc = decode_v1.postprocess_multi(pred, BODY_PARTS, POSE_PAIRS, MAPIDX, W, H, 0.1,
                                     return_intermediates=True, blobs="cc")
    assert cc[0].shape != ref[0].shape


def test_literal_contours_crowd(engine, blob_walk):
    """60 people at 656x368: merged blobs with several runs per row on some parts."""
    preds = np.stack([synth.synth_net_output(4, 60, COCO_18, 46, 82), synth.synth_net_output(5, 45, COCO_18, 46, 82)])
    engine.set_net_out(preds)
    got = engine.decode_v1(BODY_PARTS, POSE_PAIRS, MAPIDX, 2, 656, 368, 0.1, kp_cap=128, person_cap=256)
    wave, group = _check_walk(engine, blob_walk, 5)
    if blob_walk == "wave":
        assert group == 0
    for b in range(2):
        _check(got[b], _oracle(preds[b:b + 1], 656, 368))


def _noise_scene(seed, parts, level, hs=46, ws=82):
    """Smoothed white noise on the first `parts` heat maps: ragged blobs with
    holes, islands in holes, spirals - whatever the threshold cuts out."""
    rng = np.random.default_rng(seed)
    f = rng.standard_normal((parts, hs, ws)).astype(np.float32)
    for _ in range(2):
        f = (np.roll(f, 1, 1) + 2 * f + np.roll(f, -1, 1)) / 4
        f = (np.roll(f, 1, 2) + 2 * f + np.roll(f, -1, 2)) / 4
    pred = np.zeros((1, 57, hs, ws), np.float32)
    pred[0, :parts] = f / f.std() * 0.2 + level
    return pred


@pytest.mark.parametrize("seed,level,W,H,hs,ws", [
    (7, 0.05, 656, 368, 46, 82), (8, 0.0, 656, 368, 46, 82), (9, 0.12, 656, 368, 46, 82), (10, 0.25, 480, 480, 46, 82),
    (11, 0.1, 1280, 720, 23, 41), (12, 0.3, 333, 247, 46, 82), (13, 0.08, 1920, 1080, 16, 28),
])
def test_literal_contours_noise(engine, seed, level, W, H, hs, ws, blob_walk):
    """Arbitrary mask topology (tens of contours per part, contours of up to a
    few thousand vertices): list order, hole borders, nesting and fill all have
    to agree with the literal walk, keypoint by keypoint."""
    pred = _noise_scene(seed, 6, level, hs, ws)   # (coarser maps at the large sizes: the 8192-run capacity)
    engine.set_net_out(pred)
    got = engine.decode_v1(BODY_PARTS, POSE_PAIRS, MAPIDX, 1, W, H, 0.1)
    _check_walk(engine, blob_walk, 6)   # (blobs with more than 32 border loops or 5 000 pieces -- a mask that is mostly
                                        #  foreground -- go through the one-group walk in both settings)
    ref = _oracle(pred, W, H)
    _check(got[0], ref)
    assert ref[0].shape[0] > 30


def test_capacity_error(engine):
    from posehip._ffi import PosehipError
    pred = np.zeros((1, 57, 46, 82), np.float32)
    pred[0, 0, 5:45:8, 5:80:10] = 0.9          # 40 isolated blobs on part 0
    engine.set_net_out(pred)
    with pytest.raises(PosehipError):
        engine.decode_v1(BODY_PARTS, POSE_PAIRS, MAPIDX, 1, 656, 368, 0.1, kp_cap=4)


def test_negative_threshold_is_rejected(engine):
    """The blob arg-max orders values by their bit patterns, so mask pixels must be positive."""
    from posehip._ffi import PosehipError
    engine.set_net_out(np.zeros((1, 57, 46, 82), np.float32))
    with pytest.raises(PosehipError):
        engine.decode_v1(BODY_PARTS, POSE_PAIRS, MAPIDX, 1, 656, 368, -0.1)
