"""CPU: oracle/movenet.py against hand-computed cases.  The reference's module imports tensorflow at
its top, so it cannot produce fixtures here; every expected value below is worked out by hand from the
reference's source lines (movenet/utils/image_sequence_crop.py) or from TensorFlow's published crop_and_resize."""
import numpy as np
import pytest

from oracle import movenet as mv


def _kps(entries):
    k = np.zeros((1, 1, 17, 3), np.float32)
    for name, (y, x, s) in entries.items():
        k[0, 0, mv.KEYPOINT_DICT[name]] = (y, x, s)
    return k


TORSO = {'left_hip': (0.5, 0.45, 0.9), 'right_hip': (0.5, 0.55, 0.9),
         'left_shoulder': (0.3, 0.45, 0.9), 'right_shoulder': (0.3, 0.55, 0.9)}


def test_init_crop_region():
    r = mv.init_crop_region(1080, 1920)                      # :22-26
    assert r['x_min'] == 0.0 and r['width'] == 1.0 and r['x_max'] == 1.0
    assert r['height'] == 1920 / 1080 and r['y_min'] == (540 - 960) / 1080 and r['y_max'] == r['y_min'] + r['height']
    r = mv.init_crop_region(1920, 1080)                      # :27-31
    assert r['y_min'] == 0.0 and r['height'] == 1.0 and r['width'] == 1920 / 1080 and r['x_min'] == (540 - 960) / 1080
    r = mv.init_crop_region(500, 500)
    assert r == {'y_min': 0.0, 'x_min': 0.0, 'y_max': 1.0, 'x_max': 1.0, 'height': 1.0, 'width': 1.0}


def test_determine_crop_region_cases():
    # centre (500, 500); torso ranges 200 / 50 -> half = max(95, 380, 240, 60) = 380 <= room 500
    r = mv.determine_crop_region(_kps(TORSO), 1000, 1000)
    for k, v in {'y_min': .12, 'x_min': .12, 'y_max': .88, 'x_max': .88, 'height': .76, 'width': .76}.items():
        assert r[k] == pytest.approx(v, abs=1e-6)
    # a confident far joint widens the body range: wrist at (0.95, 0.5): |500-950| * 1.2 = 540 > room 500 -> half = 500,
    # not > max(w, h)/2 = 500 -> crop is the whole image
    far = dict(TORSO, left_wrist=(0.95, 0.5, 0.9))
    r = mv.determine_crop_region(_kps(far), 1000, 1000)
    for k, v in {'y_min': 0, 'x_min': 0, 'y_max': 1, 'x_max': 1, 'height': 1, 'width': 1}.items():
        assert r[k] == pytest.approx(v, abs=1e-6)
    # the same joint below the confidence floor is ignored (:81-82)
    low = dict(TORSO, left_wrist=(0.95, 0.5, 0.19))
    assert mv.determine_crop_region(_kps(low), 1000, 1000)['height'] == pytest.approx(.76, abs=1e-6)
    # torso not visible (:43-56): both hips below the floor -> default crop.  (A float32 score of 0.2 is
    # 0.2000000030 and passes the reference's `> 0.2` under float64 comparison.)
    nohips = dict(TORSO, left_hip=(0.5, 0.45, 0.1), right_hip=(0.5, 0.55, 0.19))
    assert mv.determine_crop_region(_kps(nohips), 720, 1280) == mv.init_crop_region(720, 1280)
    # crop larger than half the long side (:135-136): wide image, centre near the left edge
    wide = {'left_hip': (0.5, 0.05, 0.9), 'right_hip': (0.5, 0.07, 0.9),
            'left_shoulder': (0.1, 0.05, 0.9), 'right_shoulder': (0.1, 0.07, 0.9), 'nose': (0.5, 0.95, 0.9)}
    assert mv.determine_crop_region(_kps(wide), 500, 2000) == mv.init_crop_region(500, 2000)


def test_crop_and_resize_bilinear_and_extrapolation():
    img = np.array([[0, 10], [20, 30]], np.uint8)[:, :, None].repeat(3, axis=2)
    full = {'y_min': 0.0, 'x_min': 0.0, 'y_max': 1.0, 'x_max': 1.0}
    out = mv.crop_and_resize(img, full, (3, 3))
    assert out.dtype == np.int32 and out.shape == (3, 3, 3)
    assert out[:, :, 0].tolist() == [[0, 5, 10], [10, 15, 20], [20, 25, 30]]
    tall = {'y_min': -0.5, 'x_min': 0.0, 'y_max': 1.5, 'x_max': 1.0}       # rows at -0.5, 0.5, 1.5: outside -> 0
    assert mv.crop_and_resize(img, tall, (3, 3))[:, :, 1].tolist() == [[0, 0, 0], [10, 15, 20], [0, 0, 0]]
    # truncation of the int32 cast: 0 + (10 - 0) * 0.25 = 2.5 -> 2
    quarter = {'y_min': 0.0, 'x_min': 0.25, 'y_max': 0.0, 'x_max': 0.25}
    assert mv.crop_and_resize(img, quarter, (1, 1))[0, 0].tolist() == [2, 2, 2]


def test_update_keypoints_maps_crop_to_image():
    region = {'y_min': .12, 'x_min': -.2, 'y_max': .88, 'x_max': 1.2, 'height': .76, 'width': 1.4}
    k = _kps({'nose': (0.5, 0.5, 0.7), 'left_ankle': (1.0, 0.0, 0.3)})
    u = mv.update_keypoints(k, region, 1000, 800)
    assert u.dtype == np.float32 and u[0, 0, 0, 2] == np.float32(0.7)
    assert u[0, 0, 0, 0] == pytest.approx(0.5, abs=1e-6) and u[0, 0, 0, 1] == pytest.approx(0.5, abs=1e-6)
    assert u[0, 0, 15, 0] == pytest.approx(0.88, abs=1e-6) and u[0, 0, 15, 1] == pytest.approx(-0.2, abs=1e-6)


def test_multipose_decode_specification_case():
    G = 8
    center = np.full((G, G), 0.05, np.float32)
    center[2, 3], center[5, 5], center[5, 6] = 0.9, 0.8, 0.8          # (5,5) and (5,6): equal neighbours, both maxima
    heat = np.full((G, G, 17), 0.1, np.float32)
    regr = np.zeros((G, G, 34), np.float32)
    offs = np.zeros((G, G, 34), np.float32)
    box = np.zeros((G, G, 2), np.float32)
    regr[2, 3, 0::2], regr[2, 3, 1::2] = 1.0, -1.0                     # every joint: one down, one left -> (3, 2)
    heat[3, 2, :] = 0.7
    offs[3, 2, 0::2], offs[3, 2, 1::2] = 0.25, 0.5
    box[2, 3] = (4.0, 2.0)
    out = mv.multipose_decode(center, heat, regr, offs, box, max_people=4, min_center=0.1)
    assert out.shape == (4, 56) and out.dtype == np.float32
    assert np.allclose(out[0, :51].reshape(17, 3), [3.25 / 8, 2.5 / 8, 0.7])
    assert out[0, 51:].tolist() == [0.0, 0.25, 0.5, 0.5, np.float32(0.9)]
    assert out[1, 55] == np.float32(0.8) and out[2, 55] == np.float32(0.8)
    # ties: lower raster index first -> (5,5) before (5,6); their joints sit on the cell nearest the centre among
    # equal heat values except the 0.7 cell: 0.7 / (d + 1.8) beats 0.1 / 1.8 while d < 10.8 -> still (3, 2)
    assert out[1, 51 + 0] == 5 / 8 and out[1, 52] == 5 / 8 and out[2, 52] == 6 / 8
    assert np.allclose(out[1, :51].reshape(17, 3)[:, 2], 0.7)
    assert not out[3].any()                                             # only three centres above min_center


def test_constants_match_the_reference_file():
    """movenet/common.py is constants only and imports here: tests/golden/callers_ref.json holds its values."""
    import json
    import os
    from posehip import movenet as pmv
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "callers_ref.json")) as f:
        ref = json.load(f)["movenet_common"]
    edges = {(a, b): c for a, b, c in ref["KEYPOINT_EDGE_INDS_TO_COLOR"]}
    for mod in (mv, pmv):
        assert mod.KEYPOINT_DICT == ref["KEYPOINT_DICT"]
        assert mod.MIN_CROP_KEYPOINT_SCORE == ref["MIN_CROP_KEYPOINT_SCORE"]
    assert pmv.MODEL_URL == ref["MODEL_URL"] and pmv.KEYPOINT_EDGE_INDS_TO_COLOR == edges
