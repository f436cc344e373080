"""GPU: the estimator classes that keep the reference's interface, end to end
through the C ABI, plus the committed golden scenes."""
import argparse
import os

import numpy as np
import pytest

from posehip import _ffi, api_native, api_opencv, caffe_io, graph, synth, tables
from posehip.tables import BODY_25, COCO_18, MAPIDX

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


@pytest.mark.parametrize("tag,model", [("b25_p3", BODY_25), ("b25_p0", BODY_25), ("b25_sq_p2", BODY_25),
                                       ("coco_p4", COCO_18)])
@pytest.mark.parametrize("connector", [0, 1])
def test_golden_v2(engine, tag, model, connector):
    g = np.load(os.path.join(G, "v2_scenes.npz"))
    sfx = "1" if connector else ""
    net = g[tag + "_net"].astype(np.float32)
    engine.set_net_out(net[None])
    res, ex = engine.decode_v2(model.with_connector(connector), 1, net.shape[1] * 8, net.shape[2] * 8,
                               want_peaks=True, want_subsets=True)
    np.testing.assert_array_equal(res[0][0], g[tag + "_kp" + sfx])
    np.testing.assert_array_equal(res[0][1], g[tag + "_sc" + sfx])
    np.testing.assert_array_equal(ex["peaks"][0, :, 0, 0].astype(np.int32), g[tag + "_peaks_n"])
    np.testing.assert_array_equal(ex["subsets"][0, :len(g[tag + "_rows" + sfx])], g[tag + "_rows" + sfx])


@pytest.mark.parametrize("tag", ["p3_480", "p4_656", "p2_1080"])
def test_golden_v1(engine, tag):
    g = np.load(os.path.join(G, "v1_scenes.npz"))
    bp, pp = tables.get_body_parts_and_pose_pairs("COCO")
    W, H = (int(v) for v in g[tag + "_size"])
    engine.set_net_out(g[tag + "_net"].astype(np.float32)[None])
    kl, people, _ = engine.decode_v1(bp, pp, MAPIDX, 1, W, H, 0.1)[0]
    np.testing.assert_array_equal(kl, g[tag + "_kl"])
    np.testing.assert_array_equal(np.stack(people) if people else np.zeros((0, 19)), g[tag + "_people"])


def test_native_estimator_end_to_end(tmp_path):
    """PyOpenPoseNative.predict / filter_prediction / predict_hm / predict_from_hm /
    predict_batch with the same call sequence as openpose/native/python/test.py."""
    params = dict(model_folder="/nonexistent/", model_pose="BODY_25", net_resolution="-1x128",
                  posehip_random_weights=0)
    pyop = api_native.PyOpenPoseNative(params, 0.0, 2, 2)
    assert pyop.params["disable_blending"] is True and pyop.params["scale_number"] == 1
    img = synth.synth_frames(0, 1, 96, 160)[0]
    pyop.predict(img)
    pyop.filter_prediction()
    assert pyop.datum.netInputSizes[0].x == 208 and pyop.datum.netInputSizes[0].y == 128   # 128*160/96/16 = 13.3 -> 13
    assert pyop.opencv_image.shape == img.shape
    if pyop.pose_empty:
        assert pyop.pose_scores is None and pyop.pose_bounding_box is None
    else:
        assert pyop.pose_keypoints.shape[1:] == (25, 3) and pyop.pose_scores.ndim == 1
        assert pyop.pose_bounding_box.shape == (len(pyop.pose_scores), 4)
        assert "/" in pyop.filtered_skel
    # raw heat maps -> .float file -> decode-only entry gives the same people
    hm_path = str(tmp_path / "hm" / "f0.float")
    pyop.predict_hm(img, hm_path)
    kp0 = None if pyop.pose_empty else pyop.pose_keypoints.copy()
    hm = api_native.read_heatmap_file(hm_path)
    assert hm[0].shape == (1, 78, 16, 26)
    pyop.predict_from_hm(img, hm)
    kp1 = None if pyop.pose_empty else pyop.pose_keypoints
    assert (kp0 is None) == (kp1 is None)
    if kp0 is not None:
        np.testing.assert_array_equal(kp0, kp1)
    # injected synthetic people: decode-only path finds them and scales to image pixels
    scene = synth.synth_net_output(5, 2, BODY_25, 46, 82)[None]
    big = np.zeros((736, 1312, 3), np.uint8)
    pyop.predict_from_hm(big, [scene])
    assert not pyop.pose_empty and len(pyop.pose_scores) >= 1
    assert pyop.pose_keypoints[..., 0].max() > 656 * 0.9      # x scaled by (1312-1)/(656-1) ~ 2
    # batch == sequential
    frames = synth.synth_frames(1, 3, 96, 160)
    batch = pyop.predict_batch(list(frames))
    for f, (kp, sc) in zip(frames, batch):
        pyop.predict(f)
        if kp is None:
            assert pyop.pose_empty
        else:
            np.testing.assert_array_equal(pyop.pose_keypoints, kp)
    pyop.opWrapper.close()
    with pytest.raises(FileNotFoundError):
        api_native.PyOpenPoseNative(dict(model_folder="/nonexistent/", model_pose="BODY_25"))


def test_extractor_and_csv(tmp_path):
    args = argparse.Namespace(op_model_folder="/nonexistent/", op_model_pose="BODY_25", op_net_resolution="-1x64",
                              op_heatmaps_add_parts=True, op_heatmaps_add_bkg=True, op_heatmaps_add_PAFs=True,
                              op_heatmaps_scale=1, op_display=1, op_skel_thres=0.0, op_max_true_body=8,
                              op_patch_offset=2, posehip_random_weights=1)
    pe = api_native.OpenPosePoseExtractor(args)
    img = synth.synth_frames(2, 1, 64, 64)[0]
    out = str(tmp_path / "skel" / "0.txt")
    pe.predict(img, None, kpt_save_path=out)
    assert os.path.exists(out)
    hm = pe.pyop.datum.poseHeatMaps
    assert hm.shape == (78, 64, 64)                                  # parts + bkg + PAFs at net-input size
    if not pe.pyop.pose_empty:
        assert pe.pyop.pose_heatmaps.shape == (64, 64, 78)            # HWC view the tracker crops (track.py)
    ok, image = pe.display("w", scale=1.0, image=img)
    assert ok and image.shape[1] == 64
    pe.pyop.opWrapper.close()


def test_opencv_estimator_end_to_end():
    """PyOpenPoseOpenCV.load / predict / postprocess_* as in openpose/opencv/inference.py:57-77."""
    from oracle import decode_v1, net_ref
    bp, pp = tables.get_body_parts_and_pose_pairs("COCO")
    pyop = api_opencv.PyOpenPoseOpenCV(64, 48, bp, pp)
    pyop.load("/nonexistent.prototxt", "/nonexistent.caffemodel", False, random_weights=3)
    img = synth.synth_frames(3, 1, 60, 80)[0]                       # resized 80x60 -> 64x48 by the preprocess kernel
    pred = pyop.predict(img)
    assert pred.shape == (1, 57, 6, 8) and pred.dtype == np.float32
    assert (pyop.original_image_width, pyop.original_image_height) == (80, 60)
    t, _ = pyop.net.getPerfProfile()
    assert t > 0
    pts = pyop.postprocess_single(pred)
    assert pts == decode_v1.postprocess_single(pred, 19, 80, 60, 0.1)
    kl, people = pyop.postprocess_multi(pred)
    rk, rp = decode_v1.postprocess_multi(pred, bp, pp, MAPIDX, 80, 60, 0.1)
    np.testing.assert_array_equal(kl, rk)
    assert len(people) == len(rp)
    # synthetic people through the same class: the decode works on any [1,57,h,w] prediction
    pyop.original_image_width, pyop.original_image_height = 480, 480
    scene = synth.synth_net_output(9, 3, COCO_18, 46, 46)[None]
    kl, people = pyop.postprocess_multi(scene)
    rk, rp = decode_v1.postprocess_multi(scene, bp, pp, MAPIDX, 480, 480, 0.1)
    np.testing.assert_array_equal(kl, rk)
    for a, b in zip(people, rp):
        np.testing.assert_array_equal(a, b)
    assert len(people) >= 2
    out = pyop.draw_skeleton(personwise_keypoints=people, keypoints_list=kl, image=np.zeros((480, 480, 3), np.uint8))
    assert out.any()
    pyop.net.engine.close()


def test_preprocess_modes_match_restated_cv2(engine):
    """K1: u8 HWC BGR -> bf16 NHWC input; integer fixed-point bilinear of the
    blobFromImage resize is bit-exact against a NumPy restatement."""
    net = caffe_io.parse_prototxt('input: "image" input_dim: 1 input_dim: 3 input_dim: 24 input_dim: 40\n' +
                                  caffe_io._conv("out", "image", 8, 3))
    plan = graph.build_plan(net, graph.random_weights(net, 0))
    engine.build(plan, 2)
    frames = synth.synth_frames(4, 2, 24, 40)
    engine.preprocess(frames, _ffi.PH_PRE_CV2)                       # same size: pure x/255
    buf = engine.get_buffer(0)                                       # [B, H+2, W+2, 8]
    want = (frames.astype(np.float32) * np.float32(1 / 255.0))
    want16 = (want.view(np.uint32) + 0x7fff + ((want.view(np.uint32) >> 16) & 1)) >> 16 << 16
    np.testing.assert_array_equal(buf[:, 1:-1, 1:-1, :3].view(np.uint32), want16.astype(np.uint32))
    assert (buf[:, 0] == 0).all() and (buf[:, :, 0] == 0).all() and (buf[..., 3:] == 0).all()   # zero border + pad
    big = synth.synth_frames(5, 2, 51, 77)
    engine.preprocess(big, _ffi.PH_PRE_CV2)
    got = engine.get_buffer(0)[:, 1:-1, 1:-1, :3]
    from oracle import preprocess as pre_ref
    ref = pre_ref.cv_resize_u8(big, 40, 24).astype(np.float32) * np.float32(1 / 255.0)
    ref16 = ((ref.view(np.uint32) + 0x7fff + ((ref.view(np.uint32) >> 16) & 1)) >> 16 << 16).astype(np.uint32)
    np.testing.assert_array_equal(got.view(np.uint32), ref16)
    engine.preprocess(big, _ffi.PH_PRE_NATIVE)                       # keep-aspect, x/256-0.5, zero fill -> -0.5
    nat = engine.get_buffer(0)[:, 1:-1, 1:-1, :3]
    want = pre_ref.cv_mat_to_op_input(big, 40, 24).transpose(0, 2, 3, 1)
    np.testing.assert_array_equal(nat, want)                         # multiples of 1/256: exact in bf16
    s = min(39 / 76, 23 / 50)
    assert (nat[:, :, int(77 * s) + 2:] == -0.5).all()               # right of the scaled image: filled with 0 -> -0.5


@pytest.mark.parametrize("src_hw,net_hw,real", [
    ((1080, 1920), (368, 656), False),     # BASELINE configs[4]: 1080p stream at "-1x368" (linear, s = 0.3401)
    ((480, 480), (368, 368), False),       # the reference's cv2 bench frame size (opencv/inference.py:67), s = 0.766
    ((384, 384), (368, 368), False),       # native/python/test.py:34-35
    ((431, 577), (368, 496), False),       # non-integer down-scale with a filled margin
    ((200, 301), (368, 560), False),       # up-scale: s = 1.836 > 1 -> INTER_CUBIC
    ((368, 656), (368, 656), False),       # same size: copy
    ((368, 368), (368, 656), True),        # data/test/pillar_text.jpg into a wider net: s = 1, zero fill
    ((0, 0), (368, 496), True),            # data/sample/COCO_val2014_000000000192.jpg (480x640), s = 0.766
])
def test_native_preprocess_matches_restated_cvMatToOpInput(engine, src_hw, net_hw, real):
    """PH_PRE_NATIVE (first kernel of the headline path and of Wrapper.emplaceAndPop) against
    oracle/preprocess.py:cv_mat_to_op_input = OpenPose resizeFixedAspectRatio (cv::warpAffine, linear for
    s <= 1, cubic for s > 1, zero fill) + x/256 - 0.5, bit-exact (integer fixed point; the values are multiples of
    1/256 in [-0.5, 0.5), exact in bf16)."""
    import os
    from oracle import preprocess as pre_ref
    Hn, Wn = net_hw
    if real:
        g = np.load(os.path.join(os.path.dirname(__file__), "golden", "real_images.npz"))
        frames = g["pillar_text_bgr" if src_hw[0] else "coco_192_bgr"][None]
    else:
        frames = synth.synth_frames(src_hw[0] * 7 + src_hw[1], 2, src_hw[0], src_hw[1])
    net = caffe_io.parse_prototxt(f'input: "image" input_dim: 1 input_dim: 3 input_dim: {Hn} input_dim: {Wn}\n' +
                                  caffe_io._conv("out", "image", 8, 3))
    engine.build(graph.build_plan(net, graph.random_weights(net, 0)), len(frames))
    engine.preprocess(frames, _ffi.PH_PRE_NATIVE)
    got = engine.get_buffer(0)
    want = pre_ref.cv_mat_to_op_input(frames, Wn, Hn).transpose(0, 2, 3, 1)
    np.testing.assert_array_equal(got[:, 1:-1, 1:-1, :3], want)
    assert (got[:, 0] == 0).all() and (got[:, :, -1] == 0).all() and (got[..., 3:] == 0).all()
    if not real:   # the same frames from device memory
        import torch
        engine.preprocess(torch.from_numpy(frames).cuda(), _ffi.PH_PRE_NATIVE)
        np.testing.assert_array_equal(engine.get_buffer(0)[:, 1:-1, 1:-1, :3], want)


def test_real_images_preprocess_matches_golden(engine):
    """G6: the reference's own JPEGs (decoded with PIL as read_color_file does, utils_rs.py:43-44)
    through the blobFromImage restatement -- committed inputs + expected u8 resize, GPU bit-exact."""
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "real_images.npz"))
    net = caffe_io.parse_prototxt('input: "image" input_dim: 1 input_dim: 3 input_dim: 368 input_dim: 656\n' +
                                  caffe_io._conv("out", "image", 8, 3))
    plan = graph.build_plan(net, graph.random_weights(net, 0))
    engine.build(plan, 1)
    for tag in ("pillar_text", "coco_192"):
        img, want_u8 = g[tag + "_bgr"], g[tag + "_resized_656x368"]
        engine.preprocess(img[None], _ffi.PH_PRE_CV2)
        got = engine.get_buffer(0)[:, 1:-1, 1:-1, :3]
        ref = want_u8[None].astype(np.float32) * np.float32(1 / 255.0)
        ref16 = ((ref.view(np.uint32) + 0x7fff + ((ref.view(np.uint32) >> 16) & 1)) >> 16 << 16).astype(np.uint32)
        np.testing.assert_array_equal(got.view(np.uint32), ref16)


def test_batch_pipeline_matches_sequential(engine):
    """Two batches in flight on two engines/streams deliver exactly what one engine delivers batch by batch."""
    import torch
    from posehip.pipeline import BatchPipeline
    from posehip.tables import BODY_25
    h, w, B = 64, 96, 3
    net = caffe_io.parse_prototxt(caffe_io.body25_prototxt(h, w))
    weights = graph.random_weights(net, seed=11)
    plan = graph.build_plan(net, weights)
    batches = [torch.from_numpy(synth.synth_frames(70 + i, B if i != 2 else 2, h, w)).cuda() for i in range(5)]
    engine.build(plan, B)
    want = []
    for f in batches:
        nb = engine.preprocess(f, _ffi.PH_PRE_NATIVE)
        engine.forward(nb)
        res, extras = engine.decode_v2(BODY_25, nb, h, w, 1.0, 128)
        want.append((extras["n_people"].copy(), res))
    pipe = BatchPipeline(plan, B, BODY_25, h, w, device=0, depth=2)
    tickets = [pipe.submit(f) for f in batches]
    for t, (n, res) in zip(reversed(tickets), reversed(want)):       # any collection order
        gn, gkp, gsc = pipe.result(t)
        np.testing.assert_array_equal(gn, n)
        for b, (kp, sc) in enumerate(res):
            np.testing.assert_array_equal(gkp[b, :len(kp)], kp)
            np.testing.assert_array_equal(gsc[b, :len(sc)], sc)
    pipe.close()
    # frames that arrive in host memory (pinned tensors and plain NumPy arrays): copy stream + two device frame
    # buffers per engine; 9 submissions on 2 engines re-use every buffer at least twice
    pipe = BatchPipeline(plan, B, BODY_25, h, w, device=0, depth=2)
    order = [0, 1, 2, 3, 4, 2, 0, 4, 1]
    host = [f.cpu().pin_memory() if i % 2 else f.cpu().numpy() for i, f in enumerate(batches)]
    tickets = [pipe.submit_host(host[i]) for i in order]
    for t, i in zip(tickets, order):
        gn, gkp, gsc = pipe.result(t)
        n, res = want[i]
        np.testing.assert_array_equal(gn, n)
        for b, (kp, sc) in enumerate(res):
            np.testing.assert_array_equal(gkp[b, :len(kp)], kp)
            np.testing.assert_array_equal(gsc[b, :len(sc)], sc)
    with pytest.raises(ValueError):
        pipe.submit_host(batches[0])              # a device tensor is not host memory
    pipe.close()


def test_three_engines_on_three_threads(engine):
    """The reference creates 3 estimator instances on 3 threads (datacollection/utils.py:17-32): contexts share
    no mutable state, so concurrent use gives exactly the single-threaded results."""
    import threading
    from posehip.engine import Engine
    from posehip.tables import BODY_25
    h, w, B = 48, 80, 2
    net = caffe_io.parse_prototxt(caffe_io.body25_prototxt(h, w))
    weights = graph.random_weights(net, seed=3)
    plan = graph.build_plan(net, weights)
    frames = [synth.synth_frames(200 + i, B, h, w) for i in range(3)]
    engine.build(plan, B)
    want = []
    for f in frames:
        engine.preprocess(f, _ffi.PH_PRE_NATIVE)
        engine.forward(B)
        want.append((engine.net_out(B).copy(), engine.decode_v2(BODY_25, B, h, w, 1.0, 128)[0]))
    got, errs = [None] * 3, []

    def work(i):
        try:
            e = Engine(0)
            e.build(plan, B)
            for _ in range(5):
                e.preprocess(frames[i], _ffi.PH_PRE_NATIVE)
                e.forward(B)
                got[i] = (e.net_out(B).copy(), e.decode_v2(BODY_25, B, h, w, 1.0, 128)[0])
            e.close()
        except Exception as ex:  # pragma: no cover
            errs.append(ex)

    ths = [threading.Thread(target=work, args=(i,)) for i in range(3)]
    [t.start() for t in ths]
    [t.join() for t in ths]
    assert not errs, errs
    for (o, r), (go, gr) in zip(want, got):
        np.testing.assert_array_equal(o, go)
        for (kp, sc), (gkp, gsc) in zip(r, gr):
            np.testing.assert_array_equal(kp, gkp)
            np.testing.assert_array_equal(sc, gsc)


def test_pipeline_callers_on_the_engine(tmp_path):
    """the reference's folder walk (inference.py:355-427), the three-thread extractor ->
    tracker -> writer (:22-165) and the batched walk on the real engine: same CSV files either way."""
    from PIL import Image
    from posehip import args as ph_args
    from posehip import inference
    frames = synth.synth_frames(7, 5, 64, 96)
    for tag in ("a", "b"):
        os.makedirs(tmp_path / tag / "color")
        for i, f in enumerate(frames):
            Image.fromarray(np.ascontiguousarray(f[:, :, ::-1])).save(str(tmp_path / tag / "color" / f"{i:06d}.png"))
    # model FILES in OpenPose's folder layout (written here: none exist offline), output convs loud enough
    # that the random net's noise assembles into "people" the writers have something to write about
    from posehip import caffe_io, graph
    proto = caffe_io.body25_prototxt(368, 656)
    mdir = tmp_path / "models" / "pose" / "body_25"
    os.makedirs(mdir)
    open(mdir / "pose_deploy.prototxt", "w").write(proto)
    caffe_io.write_caffemodel(str(mdir / "pose_iter_584000.caffemodel"),
                              graph.random_weights(caffe_io.parse_prototxt(proto), seed=3, out_gain=1.0))
    ns, _ = ph_args.get_parser().parse_known_args(
        ["--op-model-folder", str(tmp_path / "models") + "/", "--op-net-resolution=-1x64", "--op-image-width", "96",
         "--op-image-height", "64", "--op-save-skel", "true", "--op-skel-thres", "0.0", "--op-display", "0",
         "--posehip-batch", "4"])
    pe = api_native.OpenPosePoseExtractor(ns)
    ns.op_input_color_image = str(tmp_path / "a" / "color")
    rt = inference.extract_2dskeletons_offline(ns, pe)
    assert len(rt["PE"]) == 5 and all(v > 0 for v in rt["PE"])
    # the reference's walk names every frame's file '<prefix>_skeleton.csv' with prefix '' (:395-404), so one
    # file survives per folder: the last frame's
    last_a = np.loadtxt(str(tmp_path / "a" / "skeleton" / "_skeleton.csv"), delimiter=",", ndmin=2)
    ns.op_input_color_image = str(tmp_path / "b" / "color")
    out = inference.extract_2dskeletons_offline_batched(ns, pe)
    assert out["frames"] == 5 and out["failed"] == 0 and len(out["filtered_skel"]) == 5
    last_b = np.loadtxt(str(tmp_path / "b" / "skeleton" / "_skeleton.csv"), delimiter=",", ndmin=2)
    assert last_a.shape == last_b.shape and np.array_equal(last_a, last_b)
    assert last_a.shape[0] >= 1 and last_a.shape[1] == 1 + 25 * 3       # score, then x,y,s per joint (:186-199)
    assert all(fs != "0" for fs in out["filtered_skel"])

    class Tracker:
        name, tracks = "t", []

        def __init__(self):
            self.n = 0

        def update(self, boxes, scores, keypoints, heatmaps, image_size):
            assert boxes.shape == (len(scores), 4) and keypoints.shape[1:] == (25, 3) and image_size == (96, 64)
            assert heatmaps.shape[:2] == (64, 96)
            self.n += 1

        def no_measurement_predict_and_update(self):
            self.n += 1

    tk = Tracker()
    rt = inference.extract_2dskeletons_and_track_offline(ns, tracker=tk, pose_extractor=pe)
    assert tk.n == 5 and len(rt["PE"]) == 5
    pe.pyop.opWrapper.close()


def test_tracker_features_device_equals_host_path():
    """create_detections on the exported pose_heatmaps (host, as the reference) and the on-device
    feature give the same detections; --heatmaps_scale 1 as OpenPosePoseExtractor passes it."""
    from posehip import utils_track
    from posehip.tables import BODY_25
    params = dict(model_folder="/nonexistent/", model_pose="BODY_25", net_resolution="-1x368", posehip_random_weights=0,
                  heatmaps_add_parts=True, heatmaps_add_bkg=True, heatmaps_add_PAFs=True, heatmaps_scale=1)
    pyop = api_native.PyOpenPoseNative(params, 0.0, 8, 2)
    img = np.zeros((368, 656, 3), np.uint8)
    pyop.predict_from_hm(img, [synth.synth_net_output(41, 4, BODY_25, 46, 82)[None]])
    assert not pyop.pose_empty and len(pyop.pose_scores) >= 3
    hm = pyop.pose_heatmaps
    assert hm.shape == (368, 656, 78) and hm.min() >= 0.0 and hm.max() <= 1.0        # mode 1: everything in [0,1]
    image_size = (656, 368)
    host = utils_track.create_detections(pyop.pose_keypoints, pyop.pose_scores, pyop.pose_bounding_box, hm,
                                         [hm.shape[1] / image_size[0], hm.shape[0] / image_size[1]])
    dev = utils_track.create_detections_device(pyop, image_size)
    assert len(host) == len(dev) == len(pyop.pose_scores)
    for h, d in zip(host, dev):
        assert h.feature.shape == d.feature.shape == (50 * 25 * 78,) and d.feature.dtype == np.float64
        np.testing.assert_array_equal(h.feature, d.feature)
        np.testing.assert_array_equal(h.tlwh, d.tlwh)
        assert h.feature.any()
    # second frame of a batch
    d2 = [api_native.Datum(), api_native.Datum()]
    for i, d in enumerate(d2):
        d.cvInputData = img
        d.customInputNetData = [synth.synth_net_output(50 + i, 2 + i, BODY_25, 46, 82)[None]]
    pyop.opWrapper.emplaceAndPop(d2)
    pyop.adopt(d2[1])
    feats = pyop.pose_track_features()
    hm1 = pyop.pose_heatmaps
    want = utils_track.create_detections(pyop.pose_keypoints, pyop.pose_scores, pyop.pose_bounding_box, hm1, [1.0, 1.0])
    assert len(want) == len(feats) > 0
    for w, f in zip(want, feats):
        np.testing.assert_array_equal(w.feature, f)
    pyop.opWrapper.close()


def test_reference_timing_harnesses_run_through_the_estimator_classes(tmp_path, capsys):
    """The reference's two timing scripts (openpose/opencv/inference.py:33-89, native/python/test.py:9-48) as
    posehip.bench_opencv.run / posehip.bench_native.test_op_runtime: they drive the estimator classes, append one
    keypoint line per iteration and print the reference's summary lines."""
    from posehip import bench_native, bench_opencv
    out = str(tmp_path / "cv")
    mean, kl, people = bench_opencv.run("/nonexistent.jpg", out, iters=3, random_weights=0, image_width=128,
                                        image_height=96)
    files = sorted(os.listdir(out))
    assert 1 <= len(files) <= 3 and all(f.startswith("kptl") and f.endswith(".txt") for f in files)
    lines = [ln for f in files for ln in open(os.path.join(out, f)).read().split("\n")[:-1]]
    assert len(lines) == 3
    assert lines[-1] == ",".join(str(v) for v in np.asarray(kl).reshape(-1).tolist())
    assert mean > 0 and kl.shape[1] == 3
    assert "Average inference time over 3 trials" in capsys.readouterr().out
    per_frame, fps, pyop = bench_native.test_op_runtime("/nonexistent.jpg", str(tmp_path / "nat"), iters=6,
                                                        net_resolution="-1x64", random_weights=0, save=True)
    text = capsys.readouterr().out
    assert "Average inference time over 3 trials" in text and "fps" in text
    assert per_frame > 0 and fps > 0
    assert pyop.pose_empty or "/" in pyop.filtered_skel
    assert os.path.exists(tmp_path / "nat" / "predictions.txt")
    pyop.opWrapper.close()


def test_native_estimator_on_a_1080p_frame_scales_keypoints_back(tmp_path):
    """BASELINE configs[4] geometry through the estimator class: a 1920x1080 frame at "-1x368" -> 656x368 net input
    (cv::warpAffine down-scale, s = 0.3401), decode at net resolution, keypoints multiplied by upstream's
    mScaleNetToOutput (2.9433, not 1/s) with the NMS offset 0.5 / scale -- equal to the oracle's decode of the same
    net output with the oracle's own scale, bit for bit."""
    from oracle import decode_v2
    params = dict(model_folder="/nonexistent/", model_pose="BODY_25", net_resolution="-1x368",
                  posehip_random_weights=3, number_people_max=512)
    pyop = api_native.PyOpenPoseNative(params, 0.0, 2, 2)
    img = synth.synth_frames(8, 1, 1080, 1920)[0]
    hm_path = str(tmp_path / "hm.float")
    pyop.predict_hm(img, hm_path)
    assert (pyop.datum.netInputSizes[0].x, pyop.datum.netInputSizes[0].y) == (656, 368)
    raw = api_native.read_heatmap_file(hm_path)[0]
    assert raw.shape == (1, 78, 46, 82)
    scale = decode_v2.scale_net_to_output(1920, 1080, 656, 368)
    assert abs(float(scale) - 1919 / 652) < 1e-6 and float(scale) == api_native.scale_net_to_output(1920, 1080, 656, 368)
    kp_o, sc_o = decode_v2.decode(raw[0], BODY_25, 368, 656, float(scale))
    if len(sc_o) == 0:
        assert pyop.pose_empty
    else:
        np.testing.assert_array_equal(pyop.pose_keypoints, kp_o)
        np.testing.assert_array_equal(pyop.pose_scores, sc_o)
        assert pyop.pose_keypoints[..., 0].max() > 656 and pyop.pose_keypoints[..., 1].max() > 368   # image pixels
    pyop.opWrapper.close()
