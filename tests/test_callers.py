"""CPU: the pipeline callers around the estimator -- flags, helpers, file readers,
extract_2dskeletons, the three-thread ExtractSkeletonAndTrack and the folder walks.

Pinned by the REFERENCE itself where it is importable offline: `tests/golden/callers_ref.*` were
produced by the reference's own args.py / utils.py (tests/golden/make_golden.py:callers).  utils_rs.py and
inference.py import cv2 and cannot run here: their counterparts are checked against the layouts the
reference's source states (file:line in each test)."""
import argparse
import json
import os
import threading

import numpy as np
import pytest

from posehip import args as ph_args
from posehip import inference, utils, utils_rs

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def ref():
    with open(os.path.join(GOLD, "callers_ref.json")) as f:
        meta = json.load(f)
    return meta, np.load(os.path.join(GOLD, "callers_ref.npz"))


def test_every_reference_flag_with_its_default_and_type(ref):
    meta, _ = ref
    ours = {a.dest: a for a in ph_args.get_parser()._actions if a.dest != "help"}
    for f in meta["flags"]:
        a = ours[f["dest"]]
        assert a.option_strings[0] == f["flag"]
        assert a.default == f["default"] and type(a.default) is type(f["default"]), f
        assert getattr(a.type, "__name__", str(a.type)) == f["type"], f
    extra = set(ours) - {f["dest"] for f in meta["flags"]}
    assert extra == {"posehip_random_weights", "posehip_device", "posehip_batch"}
    ns, rest = ph_args.get_parser().parse_known_args(["--op-mode", "offline", "--op-save-skel", "yes", "--unknown", "1"])
    assert ns.op_mode == "offline" and ns.op_save_skel is True and rest == ["--unknown", "1"]


def test_str2bool_and_colours_match_the_reference(ref):
    meta, _ = ref
    for v, want in meta["str2bool"].items():
        if isinstance(want, bool):
            assert utils.str2bool(v) is want
        else:
            with pytest.raises(argparse.ArgumentTypeError):
                utils.str2bool(v)
    assert [list(utils.get_color(i)) for i in range(12)] == meta["get_color"]
    assert [list(utils.get_color_fast(i)) for i in range(12)] == meta["get_color_fast"]


def test_resize_tensor_bit_exact_vs_reference(ref):
    _, g = ref
    for i in range(5):
        x, y = (int(v) for v in g[f"rt{i}_xy"])
        out = utils.resize_tensor(g[f"rt{i}_in"], x, y)
        want = g[f"rt{i}_out"]
        assert out.dtype == want.dtype == np.float64 and out.shape == want.shape
        assert np.array_equal(out, want), i
    # a one-row / one-column / empty crop has no index 1: the reference's loop raises IndexError
    for shape in ((1, 5, 2), (5, 1, 2), (0, 4, 1)):
        with pytest.raises(IndexError):
            utils.resize_tensor(np.zeros(shape), 25, 50)


def test_cosine_distance_vs_reference(ref):
    _, g = ref
    assert np.array_equal(utils.cosine_distance(g["cos_a"], g["cos_b"]), g["cos_d"])
    a = g["cos_a"] / np.linalg.norm(g["cos_a"], axis=1, keepdims=True)
    b = g["cos_b"] / np.linalg.norm(g["cos_b"], axis=1, keepdims=True)
    assert np.array_equal(utils.cosine_distance(a, b, True), g["cos_d_normalized"])


def test_timer_error_dict_check(capsys):
    with utils.Timer("x", enable=True, printout=True) as t:
        pass
    assert t.duration is not None and t.duration >= 0
    assert "[INFO] : Timer >>> x" in capsys.readouterr().out
    with utils.Timer("y", enable=False) as t2:
        pass
    assert t2.duration is None                       # utils.py:30-37: nothing measured when disabled
    e1, e2 = utils.Error(), utils.Error()
    assert not utils.dict_check({"a": e1, "b": e2})
    e1.state = e2.state = True
    e1.counter = 3
    assert utils.dict_check({"a": e1, "b": e2})
    e1.reset()
    assert e1.state is False and e1.counter == 0


def test_prepare_save_paths_layout(tmp_path):
    # utils_rs.py:126-154
    d = str(tmp_path / "skeleton")
    p = utils_rs.prepare_save_paths(d, "000123", True, True, True)
    assert os.path.isdir(d)
    assert p == {"skel_save_path": os.path.join(d, "000123_skeleton.csv"),
                 "3dskel_save_path": os.path.join(d, "000123_skeleton3d.csv"),
                 "skel_image_save_path": os.path.join(d, "000123_skeleton.jpg")}
    assert utils_rs.prepare_save_paths(d, "p") == {"skel_save_path": None, "3dskel_save_path": None,
                                                   "skel_image_save_path": None}
    assert utils_rs.prepare_save_paths(None) == {"skel_save_path": None, "3dskel_save_path": None,
                                                 "skel_image_save_path": None}
    with pytest.raises(AssertionError):
        utils_rs.prepare_save_paths(None, "", True)


def test_readers(tmp_path):
    from PIL import Image
    rng = np.random.default_rng(5)
    rgb = rng.integers(0, 256, (6, 8, 3), dtype=np.uint8)
    png = str(tmp_path / "a.png")
    Image.fromarray(rgb).save(png)
    assert np.array_equal(utils_rs.read_color_file(png), rgb[:, :, ::-1])          # utils_rs.py:43-44: BGR
    raw = str(tmp_path / "a.bin")
    rgb.tofile(raw)
    flat = utils_rs.read_color_file(raw, "BGR8")
    assert flat.shape == (6 * 8 * 3,) and np.array_equal(flat.reshape(6, 8, 3), rgb)
    npy = str(tmp_path / "a.npy")
    np.save(npy, rgb)
    assert np.array_equal(utils_rs.read_color_file(npy), rgb)
    np.save(npy, rgb.astype(np.float32))
    with pytest.raises(ValueError):
        utils_rs.read_color_file(npy)
    depth = rng.integers(0, 5000, (6, 8), dtype=np.uint16)
    dp = str(tmp_path / "d.bin")
    depth.tofile(dp)
    assert np.array_equal(utils_rs.read_depth_file(dp), depth.reshape(-1))
    # sensor tree: base/device/timestamp/sensor
    for dev in ("dev2", "dev1"):
        for ts in ("200", "100"):
            os.makedirs(tmp_path / "rs" / dev / ts / "color")
    tree = utils_rs.get_rs_sensor_dir(str(tmp_path / "rs"), "color")
    assert list(tree) == ["dev1", "dev2"] and list(tree["dev1"]) == ["100", "200"]
    assert tree["dev2"]["100"] == os.path.join(str(tmp_path / "rs"), "dev2", "100", "color")


def test_yuyv_known_answers():
    # integer BT.601 (OpenCV's 20-bit constants), hand-computed: grey 128 -> 130, 235 -> 255, 16 -> 0;
    # Y=81 U=90 V=240 (red in studio range) -> B,G,R = 0,0,254 (B and G saturate at 0)
    def word(y, c):
        return np.uint16(y | (c << 8))
    d = np.array([word(128, 128), word(128, 128), word(235, 128), word(16, 128), word(81, 90), word(81, 240)])
    bgr = utils_rs._get_brg_from_yuv(d)
    assert bgr.dtype == np.uint8 and bgr.shape == (6, 3)
    assert bgr[0].tolist() == [130, 130, 130] and bgr[1].tolist() == [130, 130, 130]
    assert bgr[2].tolist() == [255, 255, 255] and bgr[3].tolist() == [0, 0, 0]
    y = (81 - 16) * 1220542 + (1 << 19)
    want = [min(255, max(0, (y + 2116026 * (90 - 128)) >> 20)),
            min(255, max(0, (y - 852492 * (240 - 128) - 409993 * (90 - 128)) >> 20)),
            min(255, max(0, (y + 1673527 * (240 - 128)) >> 20))]
    assert bgr[4].tolist() == want == bgr[5].tolist() and want == [0, 0, 254]
    with pytest.raises(ValueError):
        utils_rs._get_brg_from_yuv(d[:3])


def test_read_calib_csv(tmp_path):
    # utils_rs.py:83-122
    rows = ["848,480,424.1,240.2,600.5,601.5,brown,0.1,0.2,0.3,0.4,0.5,bgr8,30",
            "640,480,320.1,239.2,380.5,381.5,brown,0,0,0,0,0,z16,15",
            "1,0,0,0,1,0,0,0,1,0.01,0.02,0.03",
            "0.001,50.0"]
    p = str(tmp_path / "calib.csv")
    open(p, "w").write("\n".join(rows) + "\n")
    c = utils_rs.read_calib_file(p)
    assert c["color"]["width"] == 848 and c["color"]["fx"] == 600.5 and c["color"]["coeffs"] == [0.1, 0.2, 0.3, 0.4, 0.5]
    assert c["color"]["format"] == "bgr8" and c["color"]["fps"] == 30 and c["depth"]["model"] == "brown"
    assert c["T_color_depth"]["rotation"] == [1, 0, 0, 0, 1, 0, 0, 0, 1]
    assert c["T_color_depth"]["translation"] == [0.01, 0.02, 0.03]
    assert c["depth"]["depth_scale"] == 0.001 and c["depth"]["depth_baseline"] == 50.0
    j = str(tmp_path / "calib.json")
    json.dump({"color": {"fx": 1.0}}, open(j, "w"))
    assert utils_rs.read_calib_file(j) == {"color": {"fx": 1.0}}


# ---------------------------------------------------------------------------------------------
class _FakePyop:
    def __init__(self):
        self.pose_empty = False
        self.filtered_skel = "0"
        self.opencv_image = np.zeros((4, 6, 3), np.uint8)
        self.pose_bounding_box = np.array([[1.0, 1.0, 3.0, 3.0]])
        self.pose_scores = np.array([0.7], np.float32)
        self.pose_keypoints = np.zeros((1, 25, 3), np.float32)
        self.pose_heatmaps = np.zeros((4, 6, 25))


class _FakeExtractor:
    """Records what the callers hand to the estimator."""

    def __init__(self, fail_on=None, empty_on=()):
        self.pyop = _FakePyop()
        self.calls = []
        self.fail_on = fail_on
        self.empty_on = set(empty_on)

    def predict(self, image, depth, kpt_save_path=None, skel_image_save_path=None):
        n = len(self.calls)
        self.calls.append(("predict", image.shape, depth, kpt_save_path, skel_image_save_path))
        if self.fail_on == n:
            raise RuntimeError("boom")
        self.pyop.pose_empty = n in self.empty_on
        self.pyop.filtered_skel = f"{0 if self.pyop.pose_empty else 1}/1"
        self.pyop.opencv_image = np.full((4, 6, 3), n, np.uint8)

    def predict_3d(self, image, depth, intr_mat, depth_scale=1e-3, kpt_save_path=None, kpt_3d_save_path=None,
                   skel_image_save_path=None):
        self.calls.append(("predict_3d", image.shape, depth, intr_mat, depth_scale, kpt_save_path, kpt_3d_save_path,
                           skel_image_save_path))

    def display(self, win_name="1", speed=1000, scale=1.0, image=None, bounding_box=False, tracks=None):
        return True, None


def _ns(**kw):
    ns, _ = ph_args.get_parser().parse_known_args([])
    ns.op_image_height, ns.op_image_width = 4, 6
    for k, v in kw.items():
        setattr(ns, k, v)
    return ns


def _frames(folder, n):
    from PIL import Image
    os.makedirs(folder)
    for i in range(n):
        Image.fromarray(np.full((4, 6, 3), i, np.uint8)).save(os.path.join(folder, f"{i:06d}.png"))


def test_extract_2dskeletons_results_and_errors(tmp_path, capsys):
    args = _ns(op_save_skel=True)
    pe = _FakeExtractor()
    skel = str(tmp_path / "skeleton")
    out = inference.extract_2dskeletons(args, pe, np.zeros(72, np.uint8), skeleton_folder=skel,
                                        skeleton_file_prefix="7", enable_timer=True)
    assert out["status"] is True and out["data_prep_time"] >= 0 and out["infer_time"] >= 0
    assert pe.calls[-1] == ("predict", (4, 6, 3), None, os.path.join(skel, "7_skeleton.csv"), None)
    # timer off: the reference's Timer leaves duration None (inference.py:280-282)
    out = inference.extract_2dskeletons(args, pe, np.zeros((4, 6, 3), np.uint8), skeleton_folder=skel)
    assert out == {"status": True, "data_prep_time": None, "infer_time": None}
    # wrong size -> reshape error -> status False, both times -1 (:237-247)
    out = inference.extract_2dskeletons(args, pe, np.zeros(10, np.uint8), skeleton_folder=skel)
    assert out == {"status": False, "data_prep_time": -1, "infer_time": -1}
    assert "Error in reshaping image" in capsys.readouterr().out
    # unreadable file (:208-215)
    out = inference.extract_2dskeletons(args, pe, str(tmp_path / "missing.png"), skeleton_folder=skel)
    assert out["status"] is False and "Error loading data" in capsys.readouterr().out
    # estimator raises (:271-278)
    out = inference.extract_2dskeletons(args, _FakeExtractor(fail_on=0), np.zeros(72, np.uint8), skeleton_folder=skel)
    assert out["status"] is False and "Error in pose extraction" in capsys.readouterr().out
    # save flags without a folder: prepare_save_paths asserts (utils_rs.py:131-135)
    with pytest.raises(AssertionError):
        inference.extract_2dskeletons(args, pe, np.zeros(72, np.uint8))
    # 3-D branch: positional hand-over exactly as inference.py:256-263 (2-D save path lands in depth_scale)
    args3 = _ns(op_extract_3d_skel=True, op_save_skel=True, op_save_3d_skel=True)
    depth = np.zeros((4, 6), np.uint16)
    inference.extract_2dskeletons(args3, pe, np.zeros(72, np.uint8), depth_src=depth, intr_mat=[1] * 9,
                                  skeleton_folder=skel, skeleton_file_prefix="9")
    name, shape, d, intr, depth_scale, kpt, kpt3d, img = pe.calls[-1]
    assert name == "predict_3d" and d is depth and intr == [1] * 9
    assert depth_scale == os.path.join(skel, "9_skeleton.csv") and kpt == os.path.join(skel, "9_skeleton3d.csv")
    assert kpt3d is None and img is None


class _FakeTracker:
    name = "faketrack"

    def __init__(self):
        self.updates, self.misses, self.tracks = [], 0, []

    def update(self, boxes, scores, keypoints, heatmaps, image_size):
        self.updates.append((boxes.shape, scores.shape, keypoints.shape, heatmaps.shape, image_size))

    def no_measurement_predict_and_update(self):
        self.misses += 1


def test_three_stage_pipeline_tuple_layouts_and_sentinel(tmp_path):
    folder = str(tmp_path / "dev" / "color")
    _frames(folder, 3)
    files = sorted(os.path.join(folder, f) for f in os.listdir(folder))
    args = _ns(op_save_track_image=True)
    pe, tk = _FakeExtractor(empty_on={1}), _FakeTracker()
    with pytest.raises(ValueError):
        inference.ExtractSkeletonAndTrack(_ns(op_proc="mp"), pe, tk, True)
    est = inference.ExtractSkeletonAndTrack(args, pe, tk, True)
    est.start()
    outs = []
    for f in files:
        est.queue_input((f, None, str(tmp_path / "dev" / "skeleton"), None, False))
        outs.append(est.queue_output())
    est.break_process_loops()
    for t in (est.p_extract, est.p_track, est.p_write):
        t.join(timeout=5)
        assert not t.is_alive()
    # (filtered_skel, prep_time, infer_time, track_time, break_loop)  inference.py:164-165
    assert [o[0] for o in outs] == ["1/1", None, "1/1"] and all(o[4] is False for o in outs)
    assert all(o[1] > 0 and o[2] > 0 for o in outs)
    assert outs[0][3] > 0 and outs[1][3] == -1                   # :137 / :143
    assert tk.updates == [((1, 4), (1,), (1, 25, 3), (4, 6, 25), (6, 4))] * 2 and tk.misses == 1
    # op_save_track_image: <root>/<sensor>_<tracker name>/<file> for frames with people (:157-162)
    saved = sorted(os.listdir(str(tmp_path / "dev" / "color_faketrack")))
    assert saved == ["000000.png", "000002.png"]
    assert threading.active_count() >= 1


def test_offline_walk_runtime_resets_after_warmup(tmp_path):
    folder = str(tmp_path / "cam" / "color")
    _frames(folder, 18)
    args = _ns(op_input_color_image=folder, op_save_skel=True, op_display=0.0)
    pe = _FakeExtractor()
    runtime = inference.extract_2dskeletons_offline(args, pe)
    assert len(pe.calls) == 18
    assert len(runtime["PE"]) == 3 and runtime["TK"] == []          # restarted at frame 15 (:392-393)
    assert pe.calls[0][3] == os.path.join(str(tmp_path / "cam" / "skeleton"), "_skeleton.csv")   # :395-397, prefix ''
    with pytest.raises(AssertionError):
        inference.extract_2dskeletons_offline(_ns(op_input_color_image=str(tmp_path / "nope")), pe)
    tk = _FakeTracker()
    rt = inference.extract_2dskeletons_and_track_offline(args, tracker=tk, pose_extractor=_FakeExtractor())
    assert len(rt["PE"]) == 3 and len(rt["TK"]) == 3 and len(tk.updates) == 18
    n0 = len(pe.calls)
    inference.extract_2dskeletons_online(args, pe, max_passes=2)
    assert len(pe.calls) - n0 == 36


def test_cv2_flavour_writers_file_layout(tmp_path):
    """save_keypoints / save_personwise_keypoints (openpose/opencv/inference.py:17-30): one appended line of
    comma-joined Python-float reprs per call."""
    from posehip import bench_opencv
    kl = np.array([[12.0, 7.0, 0.8125], [300.0, 41.0, 0.25]])
    p = str(tmp_path / "kptl.txt")
    bench_opencv.save_keypoints(kl, p)
    bench_opencv.save_keypoints(np.zeros((0, 3)), p)                       # nobody: an empty line
    bench_opencv.save_keypoints(kl[:1], p)
    assert open(p).read() == "12.0,7.0,0.8125,300.0,41.0,0.25\n\n12.0,7.0,0.8125\n"
    # exactly what the reference's expression produces
    want = ",".join([str(i) for keypoint in kl.tolist() for i in keypoint])
    assert open(p).read().split("\n")[0] == want
    people = [np.array([0., 1., -1., 2.5]), np.array([3., -1., 4., 1.75])]
    q = str(tmp_path / "people.txt")
    bench_opencv.save_personwise_keypoints(people, q)
    bench_opencv.save_personwise_keypoints([], q)
    assert open(q).read() == "0.0,1.0,-1.0,2.5,3.0,-1.0,4.0,1.75\n\n"
    assert open(q).read().split("\n")[0] == ",".join([str(i) for k in people for i in k.tolist()])
    # the reference's import lines resolve through compat/
    import importlib
    import os
    import sys
    compat = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "compat")
    sys.path.insert(0, compat)
    try:
        for name in list(sys.modules):
            if name == "openpose" or name.startswith("openpose."):
                del sys.modules[name]
        m = importlib.import_module("openpose.opencv.inference")
        assert m.save_keypoints is bench_opencv.save_keypoints
        t = importlib.import_module("openpose.native.python.test")
        assert callable(t.test_op_runtime)
    finally:
        sys.path.remove(compat)
    # frame loader: BGR u8 of the asked size, from the reference's JPEG when it is there, seeded noise otherwise
    f = bench_opencv.load_frame("/nonexistent.jpg", 64)
    assert f.shape == (64, 64, 3) and f.dtype == np.uint8
    np.testing.assert_array_equal(f, bench_opencv.load_frame("", 64))
