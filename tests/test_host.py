"""CPU: host logic -- Caffe file I/O, execution-plan builder, C-ABI symbol
coverage, estimator-interface helpers.  No compute calls (no GPU here)."""
import argparse
import ctypes
import os
import re

import numpy as np
import pytest

from posehip import _ffi, api_native, api_opencv, caffe_io, graph, tables

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "posehip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(ph_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(_ffi.SYMBOLS), declared ^ set(_ffi.SYMBOLS)
    assert os.path.exists(_ffi.LIB_PATH), "run `make -C pose-estimation_amd` (or __graft_entry__.build())"
    lib = ctypes.CDLL(_ffi.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), name
    assert ctypes.sizeof(_ffi.PhV2Model) == 4 * 12 + 4 * 256
    assert ctypes.sizeof(_ffi.PhV1Model) == 8 + 4 * 128


def test_missing_library_fails_loudly(monkeypatch):
    monkeypatch.setattr(_ffi, "_lib", None)
    monkeypatch.setattr(_ffi, "LIB_PATH", "/nonexistent/libposehip.so")
    with pytest.raises(_ffi.PosehipError):
        _ffi.lib()


def test_prototxt_parser():
    txt = '''
    name: "t"  # comment
    input: "image" input_dim: 1 input_dim: 3 input_dim: 16 input_dim: 24
    layer { name: "c1" type: "Convolution" bottom: "image" top: "c1"
            param { lr_mult: 1.0 } convolution_param { num_output: 8 pad: 1 kernel_size: 3
            weight_filler { type: "gaussian" std: 0.01 } } }
    layer { name: "r1" type: "ReLU" bottom: "c1" top: "c1" }
    layer { name: "p1" type: "Pooling" bottom: "c1" top: "p1" pooling_param { pool: MAX kernel_size: 2 stride: 2 } }
    '''
    net = caffe_io.parse_prototxt(txt)
    assert net["name"] == ["t"] and net["input_dim"] == [1, 3, 16, 24]
    assert [l["type"][0] for l in net["layer"]] == ["Convolution", "ReLU", "Pooling"]
    assert net["layer"][0]["convolution_param"][0]["weight_filler"][0]["std"] == [0.01]
    assert net["layer"][2]["pooling_param"][0]["pool"] == ["MAX"]
    with pytest.raises(ValueError):
        caffe_io.parse_prototxt("layer { name: 'x' ")


def test_caffemodel_roundtrip(tmp_path):
    rng = np.random.default_rng(0)
    layers = {"conv1_1": [rng.normal(size=(4, 3, 3, 3)).astype(np.float32), rng.normal(size=(4,)).astype(np.float32)],
              "prelu": [rng.normal(size=(4,)).astype(np.float32)]}
    p = str(tmp_path / "m.caffemodel")
    caffe_io.write_caffemodel(p, layers)
    back = caffe_io.read_caffemodel(p)
    assert set(back) == set(layers)
    for k in layers:
        for a, b in zip(layers[k], back[k]):
            assert a.shape == b.shape and np.array_equal(a, b)


def test_body25_plan_matches_survey_numbers():
    net = caffe_io.parse_prototxt(caffe_io.body25_prototxt(368, 656))
    nodes, inp, out = graph._build_nodes(net, None, None)
    convs = [n for n in nodes if n.kind == "conv"]
    assert len(convs) == 12 + 6 * 17 and out.cout == 78 and (out.h, out.w) == (46, 82)
    w = {n.name: [np.zeros((n.cout, n.src[0].cout, n.k, n.k), np.float32), np.zeros(n.cout, np.float32)] for n in convs}
    for n in convs:
        if n.act == _ffi.PH_ACT_PRELU:
            w[n.act_name] = [np.zeros(n.cout, np.float32)]
    plan = graph.build_plan(net, w)
    assert abs(plan.flops_per_frame() / 1e9 - 287.26) < 0.05            # ~287 GFLOP/frame
    assert plan.out_channels == 78 and (plan.in_h, plan.in_w) == (368, 656)
    widths = sorted({o["cin_read"] for o in plan.ops if o["type"] == _ffi.PH_OP_CONV})
    assert widths == [8, 64, 96, 128, 192, 224, 256, 288, 384, 512]     # 3->8; concat buffers: 8-aligned members, 32-aligned total
    for o in plan.ops:
        if o["type"] != _ffi.PH_OP_CONV:
            continue
        assert o["cin_read"] % 8 == 0 and o["in_coff"] % 8 == 0 and len(o["dst"]) <= _ffi.PH_MAX_DST
        m = o["cin_map"]
        assert sorted(m[m >= 0]) == list(range(o["weight"].shape[1]))   # every Caffe input channel mapped once
        assert all(c % 8 == 0 for _, c in o["dst"])
    outs = {o["name"]: o["f32_coff"] for o in plan.ops if o["f32_coff"] >= 0}
    assert outs == {"Mconv7_stage1_L1": 0, "Mconv7_stage3_L2": 26}      # heat maps first, then PAFs
    feat = next(o for o in plan.ops if o["name"] == "conv4_4_CPM")
    assert len(feat["dst"]) == 5                                          # written once into every consumer concat
    for b in plan.bufs:
        assert b["channels"] % 8 == 0


def test_coco_plan_and_small_resolution():
    net = caffe_io.parse_prototxt(caffe_io.coco_prototxt(368, 368))
    w = graph.random_weights(net, seed=0)
    plan = graph.build_plan(net, w, height=184, width=200)
    assert plan.out_channels == 57 and (plan.out_h, plan.out_w) == (23, 25)
    assert max(o["k"] for o in plan.ops) == 7
    assert max(b["pad"] for b in plan.bufs) == 3
    outs = {o["name"]: o["f32_coff"] for o in plan.ops if o["f32_coff"] >= 0}
    assert outs == {"Mconv7_stage6_L2": 0, "Mconv7_stage6_L1": 19}


def test_unsupported_graphs_are_rejected():
    bad = caffe_io.parse_prototxt('input: "image" input_dim: 1 input_dim: 3 input_dim: 8 input_dim: 8\n'
                                  'layer { name: "c" type: "Convolution" bottom: "image" top: "c" '
                                  'convolution_param { num_output: 8 kernel_size: 3 stride: 2 pad: 1 } }')
    with pytest.raises(ValueError):
        graph.build_plan(bad, {})
    bad = caffe_io.parse_prototxt('input: "image" input_dim: 1 input_dim: 3 input_dim: 8 input_dim: 8\n'
                                  'layer { name: "s" type: "Softmax" bottom: "image" top: "s" }')
    with pytest.raises(ValueError):
        graph.build_plan(bad, {})
    # a pooled Concat and a Concat of Concats have no buffer layout here: refused with a message, not an IndexError
    head = ('input: "image" input_dim: 1 input_dim: 3 input_dim: 8 input_dim: 8\n' + caffe_io._conv("a", "image", 8, 3) +
            caffe_io._conv("b", "a", 8, 3) + caffe_io._concat("cat", ["a", "b"]))
    for tail in (caffe_io._pool("p", "cat") + caffe_io._conv("out", "p", 8, 1),
                 caffe_io._conv("c", "cat", 8, 3) + caffe_io._concat("cat2", ["cat", "c"]) + caffe_io._conv("out", "cat2", 8, 1)):
        net = caffe_io.parse_prototxt(head + tail)
        with pytest.raises(ValueError):
            graph.build_plan(net, graph.random_weights(net, 0))


def test_net_input_size():
    assert api_native.net_input_size("-1x368", 1920, 1080) == (656, 368)   # 654.2 -> 656
    assert api_native.net_input_size("-1x368", 384, 384) == (368, 368)
    assert api_native.net_input_size("656x368", 100, 100) == (656, 368)
    assert api_native.net_input_size("320x-1", 640, 480) == (320, 240)
    assert api_native.net_input_size("-1x368", 497, 368) == (496, 368)     # 31.06 sixteenths: rounded, not ceiled
    from oracle import preprocess as pre_ref
    for wh in ((1920, 1080), (640, 480), (497, 368), (1280, 720), (333, 500)):
        assert api_native.net_input_size("-1x368", *wh) == pre_ref.op_net_input_size("-1x368", *wh)
    with pytest.raises(ValueError):
        api_native.net_input_size("-1x-1", 10, 10)
    with pytest.raises(ValueError):
        api_native.net_input_size("100x100", 10, 10)


def _base_with(kp, sc):
    b = api_native.PyOpenPoseNativeBase(0.5, 2, 2)
    b.datum.poseKeypoints = kp
    b.datum.poseScores = sc
    return b


def test_native_properties_bbox_and_csv(tmp_path):
    kp = np.zeros((2, 25, 3), np.float32)
    kp[0, [1, 2, 5], :] = [[10, 20, .9], [5, 25, .8], [15.5, 22, .7]]
    kp[1, 8] = [7, 7, .5]                                      # single torso joint -> degenerate box grows by 1
    b = _base_with(kp, np.float32([.6, .2]))
    np.testing.assert_array_equal(b.pose_bounding_box, [[5, 20, 15.5, 25], [7, 7, 8, 8]])
    np.testing.assert_array_equal(b.pose_bounding_box_int, [[5, 20, 16, 25], [7, 7, 8, 8]])
    none = _base_with(np.zeros((1, 25, 3), np.float32), np.float32([.1]))
    np.testing.assert_array_equal(none.pose_bounding_box, [[0, 0, 1, 1]])
    p = str(tmp_path / "out" / "k.txt")
    b.save_pose_keypoints(p)
    rows = np.loadtxt(p, delimiter=",")
    assert rows.shape == (2, 76) and rows[0, 0] == np.float32(.6)         # [score, 25 x (x, y, s)]
    _base_with(None, None).save_pose_keypoints(str(tmp_path / "out" / "e.txt"))
    assert os.path.getsize(tmp_path / "out" / "e.txt") == 0                # nobody -> empty file
    ref = np.loadtxt(os.path.join(os.path.dirname(__file__), "golden", "ref_predictions_native.txt"), delimiter=",")
    assert ref.size == rows.shape[1]                                        # same layout as the reference's file


def test_get_3d_skeleton_matches_reference_formula():
    depth = np.full((100, 120), 2000.0, np.float32)
    depth[40:44, 50:54] = np.nan
    skel = np.float32([[60, 50, 1], [52, 42, 1], [0, 0, 0]])
    out = api_native.PyOpenPoseNativeBase.get_3d_skeleton(skel, depth, [600., 0, 60., 0, 600., 50.], 1e-3, 2)
    np.testing.assert_allclose(out[0], [0, 0, 2.0])
    assert np.isnan(out[1]).all()                                            # all-NaN patch -> nanmean NaN, as reference
    np.testing.assert_allclose(out[2], [(0 - 60) / 600 * 2, (0 - 50) / 600 * 2, 2.0], rtol=1e-6)


def test_heatmap_file_format(tmp_path):
    raw = np.arange(2 * 3 * 4, dtype=np.float32).reshape(1, 2, 3, 4)
    p = str(tmp_path / "hm.float")
    np.insert(raw.reshape(-1), [0, 0, 0, 0, 0], [raw.ndim] + list(raw.shape)).tofile(p)   # skeleton.py:460-464
    back = api_native.read_heatmap_file(p)
    assert len(back) == 1 and np.array_equal(back[0], raw)
    hdr = np.fromfile(p, np.float32)[:5]
    np.testing.assert_array_equal(hdr, [4, 1, 2, 3, 4])                     # header stored as float32


def test_extractor_reads_reference_flag_names(monkeypatch):
    seen = {}

    class FakeNative:
        def __init__(self, params, a, b, c):
            seen.update(params=params, rest=(a, b, c))
    monkeypatch.setattr(api_native, "PyOpenPoseNative", FakeNative)
    args = argparse.Namespace(op_model_folder="/m", op_model_pose="BODY_25", op_net_resolution="-1x368",
                              op_heatmaps_add_parts=False, op_heatmaps_add_bkg=False, op_heatmaps_add_PAFs=False,
                              op_heatmaps_scale=1, op_display=0, op_skel_thres=0.5, op_max_true_body=8,
                              op_patch_offset=2)
    api_native.OpenPosePoseExtractor(args)
    assert seen["params"]["model_folder"] == "/m" and seen["params"]["render_pose"] == 0
    assert seen["rest"] == (0.5, 8, 2)


def test_opencv_class_surface():
    bp, pp = tables.get_body_parts_and_pose_pairs("COCO")
    o = api_opencv.PyOpenPoseOpenCV(368, 368, bp, pp)
    assert (o.image_width, o.image_height, o.conf_thres, o.net) == (368, 368, 0.1, None)
    for name in ("load", "predict", "postprocess_single", "postprocess_multi", "draw_skeleton"):
        assert callable(getattr(o, name))
    with pytest.raises(FileNotFoundError):
        o.load("/nonexistent.prototxt", "/nonexistent.caffemodel", False)
    assert o.draw_skeleton(image=None) is None


def test_compat_import_paths_resolve_to_posehip():
    """the reference's import lines work against compat/ and land on the posehip classes."""
    import importlib
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "compat"))
    try:
        for m in [k for k in sys.modules if k == "openpose" or k.startswith("openpose.")]:
            del sys.modules[m]
        cv = importlib.import_module("openpose.opencv")
        nat = importlib.import_module("openpose.native.python.skeleton")
        from posehip import api_native, api_opencv
        assert cv.PyOpenPoseOpenCV is api_opencv.PyOpenPoseOpenCV
        assert cv.get_body_parts_and_pose_pairs("COCO")[0]["Nose"] == 0
        assert nat.PyOpenPoseNative is api_native.PyOpenPoseNative and nat.TORSO_IDS == [1, 2, 5, 8, 9, 12]
        assert nat.OpenPosePoseExtractor is api_native.OpenPosePoseExtractor
        # the pipeline callers and their helpers (openpose/native/python/{inference,args,utils,utils_rs}.py)
        from posehip import args as ph_args, inference as ph_inf, utils as ph_utils, utils_rs as ph_rs
        inf = importlib.import_module("openpose.native.python.inference")
        assert inf.extract_2dskeletons is ph_inf.extract_2dskeletons
        assert inf.ExtractSkeletonAndTrack is ph_inf.ExtractSkeletonAndTrack
        assert importlib.import_module("openpose.native.python.args").get_parser is ph_args.get_parser
        assert importlib.import_module("openpose.native.python.utils").Timer is ph_utils.Timer
        assert importlib.import_module("openpose.native.python.utils_rs").read_color_file is ph_rs.read_color_file
    finally:
        sys.path.remove(os.path.join(root, "compat"))


def test_product_path_never_touches_the_oracle():
    """oracle/ is test infrastructure: nothing under pose-estimation_amd/ or compat/ may import, open or
    execute it (only tests/, smoke() and bench.py's cpu_baseline leg do)."""
    import ast
    hits = []
    for top in ("pose-estimation_amd", "compat"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, top)):
            for fn in files:
                if not fn.endswith(".py"):
                    continue
                path = os.path.join(dirpath, fn)
                tree = ast.parse(open(path).read(), path)
                for node in ast.walk(tree):
                    names = []
                    if isinstance(node, ast.Import):
                        names = [a.name for a in node.names]
                    elif isinstance(node, ast.ImportFrom):
                        names = [node.module or ""]
                    elif isinstance(node, ast.Call) and getattr(node.func, "id", getattr(node.func, "attr", "")) in (
                            "import_module", "__import__", "exec", "eval"):
                        names = [a.value for a in node.args if isinstance(a, ast.Constant) and isinstance(a.value, str)]
                    hits += [(path, n) for n in names if n == "oracle" or n.startswith("oracle.")]
    assert not hits, hits
    for dirpath, _, files in os.walk(os.path.join(ROOT, "pose-estimation_amd", "csrc")):
        for fn in files:
            assert "#include \"../../oracle" not in open(os.path.join(dirpath, fn)).read()


def test_real_engine_comparison_function():
    """posehip/real_engine.py: the array comparison the opportunistic cv2.dnn check reports (bench.py `real_engine`);
    `run` itself needs cv2 + the Caffe model and returns None without them."""
    from posehip import real_engine
    ref = np.zeros((1, 3, 4, 5), np.float32)
    got = ref.copy()
    got[0, 1, 2, 3] = 0.004
    ref_kp = np.array([[10, 10, 0.9], [50, 20, 0.8], [200, 100, 0.7]])
    got_kp = np.array([[51, 20, 0.8], [10, 13, 0.9], [90, 90, 0.5]])
    out = real_engine.compare(ref, got, ref_kp, got_kp)
    assert abs(out["heatmap_max_abs_err"] - 0.004) < 1e-9
    assert (out["keypoints_ref"], out["keypoints_got"], out["matched"]) == (3, 3, 2)
    assert out["keypoint_max_px_err"] == 3.0
    empty = real_engine.compare(ref, ref, np.zeros((0, 3)), np.zeros((0, 3)))
    assert empty["matched"] == 0 and empty["keypoint_max_px_err"] is None and empty["heatmap_max_abs_err"] == 0.0
    assert real_engine.run("/nonexistent", "/nonexistent/*.jpg") is None


def test_ensure_hw_queues_only_raises(monkeypatch):
    """One hardware queue per stream for the host-frame pipeline (posehip/pipeline.py): the variable is raised to the
    stream count, capped, and never lowered."""
    from posehip.pipeline import ensure_hw_queues
    monkeypatch.delenv("GPU_MAX_HW_QUEUES", raising=False)
    assert ensure_hw_queues(3) == 4 and "GPU_MAX_HW_QUEUES" not in os.environ      # HIP's default already covers it
    assert ensure_hw_queues(5) == 5 and os.environ["GPU_MAX_HW_QUEUES"] == "5"
    assert ensure_hw_queues(40) == 8 and os.environ["GPU_MAX_HW_QUEUES"] == "8"    # capped
    monkeypatch.setenv("GPU_MAX_HW_QUEUES", "16")
    assert ensure_hw_queues(5) == 16 and os.environ["GPU_MAX_HW_QUEUES"] == "16"   # never lowered
