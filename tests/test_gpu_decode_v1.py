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
    return decode_v1.postprocess_multi(pred, BODY_PARTS, POSE_PAIRS, MAPIDX, W, H, thr, return_intermediates=True)


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
    (46, 82, 300, 200, 0),      # empty scene: every limb "No Connection"
])
def test_decode_v1_matches_oracle(engine, hs, ws, W, H, people):
    pred = synth.synth_net_output(40 + people, people, COCO_18, hs, ws)[None]
    engine.set_net_out(pred)
    got = engine.decode_v1(BODY_PARTS, POSE_PAIRS, MAPIDX, 1, W, H, 0.1)
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


def test_capacity_error(engine):
    from posehip._ffi import PosehipError
    pred = np.zeros((1, 57, 46, 82), np.float32)
    pred[0, 0, 5:45:8, 5:80:10] = 0.9          # 40 isolated blobs on part 0
    engine.set_net_out(pred)
    with pytest.raises(PosehipError):
        engine.decode_v1(BODY_PARTS, POSE_PAIRS, MAPIDX, 1, 656, 368, 0.1, kp_cap=4)
