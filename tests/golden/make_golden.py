#!/usr/bin/env python3
"""Regenerates the committed golden fixtures (run in the BUILD container only).

  tables_ref.json      COLORS / MAPIDX / COCO+MPI body_parts and pose_pairs read
                       from the REFERENCE file openpose/opencv/common.py (pure
                       constants; loaded by path so the package __init__, which
                       imports cv2, is not executed).
  v2_scenes.npz        seeded synthetic BODY_25 / COCO net outputs and the V2
  v1_scenes.npz        / V1 oracle's outputs for them (inputs + expected outputs;
                       produced by oracle/*.py, NOT by the reference, which cannot
                       run offline -- SURVEY 8c).
  real_images.npz      G6: data/test/pillar_text.jpg and data/sample/COCO_val2014_000000000192.jpg
                       decoded with PIL exactly as read_color_file does (utils_rs.py:43-44: RGB -> BGR),
                       plus the oracle's cv2.resize restatement of each at 656x368 (data only).
  callers_ref.json / callers_ref.npz   SURVEY 8f-1/8f-2 host helpers, produced BY THE REFERENCE's own
                       openpose/native/python/{args,utils}.py (argparse/NumPy only; imported from
                       /root/reference as a path-only package so the package __init__ is not run):
                       every CLI flag's dest/default/type name, str2bool's truth table, get_color(_fast),
                       and resize_tensor / cosine_distance outputs on seeded inputs.
  ref_predictions_native.txt  is a verbatim copy of the reference's only stored
                       result, data/test/output/inference_native/predictions.txt.
"""
import importlib.util
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "pose-estimation_amd"))


def tables():
    path = "/root/reference/openpose/opencv/common.py"
    spec = importlib.util.spec_from_file_location("ref_common", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    out = {"COLORS": mod.COLORS, "MAPIDX": mod.MAPIDX}
    for ds in ("COCO", "MPI"):
        bp, pp = mod.get_body_parts_and_pose_pairs(ds)
        out[ds] = {"body_parts": bp, "pose_pairs": pp}
    with open(os.path.join(HERE, "tables_ref.json"), "w") as f:
        json.dump(out, f, indent=1)


def v2():
    from oracle import decode_v2
    from posehip import synth
    from posehip.tables import BODY_25, COCO_18
    out = {}
    for tag, model, hs, ws, P, seed in (("b25_p3", BODY_25, 46, 82, 3, 501), ("b25_p0", BODY_25, 46, 82, 0, 502),
                                         ("b25_sq_p2", BODY_25, 46, 46, 2, 503), ("coco_p4", COCO_18, 46, 82, 4, 504)):
        net = synth.synth_net_output(seed, P, model, hs, ws)
        kp, sc, inter = decode_v2.decode(net, model, hs * 8, ws * 8, 1.0, return_intermediates=True)
        out[tag + "_net"] = net.astype(np.float16)          # halves the fixture; decoded from this exact input
        net = out[tag + "_net"].astype(np.float32)
        for suffix, connector in (("", 0), ("1", 1)):     # CPU-build / CUDA-build assembly of upstream
            kp, sc, inter = decode_v2.decode(net, model, hs * 8, ws * 8, 1.0, return_intermediates=True,
                                             connector=connector)
            out[tag + "_kp" + suffix], out[tag + "_sc" + suffix] = kp, sc
            out[tag + "_peaks_n"] = inter["peaks"][:, 0, 0].astype(np.int32)
            rows = inter["rows"][inter["valid"]] if len(inter["valid"]) else np.zeros((0, model.n_parts + 1), np.int32)
            out[tag + "_rows" + suffix] = rows
    np.savez_compressed(os.path.join(HERE, "v2_scenes.npz"), **out)


def real_images():
    from PIL import Image
    from oracle import preprocess
    out = {}
    for tag, path in (("pillar_text", "/root/reference/data/test/pillar_text.jpg"),
                      ("coco_192", "/root/reference/data/sample/COCO_val2014_000000000192.jpg")):
        bgr = np.array(Image.open(path))[:, :, ::-1].copy()
        out[tag + "_bgr"] = bgr
        out[tag + "_resized_656x368"] = preprocess.cv_resize_u8(bgr[None], 656, 368)[0]
    np.savez_compressed(os.path.join(HERE, "real_images.npz"), **out)


def v1():
    from oracle import decode_v1
    from posehip import synth
    from posehip.tables import COCO_18, MAPIDX, get_body_parts_and_pose_pairs
    bp, pp = get_body_parts_and_pose_pairs("COCO")
    out = {}
    for tag, hs, ws, W, H, P, seed in (("p3_480", 46, 46, 480, 480, 3, 601), ("p4_656", 46, 82, 656, 368, 4, 602),
                                        ("p2_1080", 46, 82, 1920, 1080, 2, 603)):
        net = synth.synth_net_output(seed, P, COCO_18, hs, ws).astype(np.float16)
        pred = net.astype(np.float32)[None]
        kl, people, _ = decode_v1.postprocess_multi(pred, bp, pp, MAPIDX, W, H, 0.1, True)
        out[tag + "_net"] = net
        out[tag + "_size"] = np.array([W, H], np.int32)
        out[tag + "_kl"] = kl
        out[tag + "_people"] = np.stack(people) if people else np.zeros((0, 19))
    np.savez_compressed(os.path.join(HERE, "v1_scenes.npz"), **out)


def callers():
    import importlib
    import types
    pkg = types.ModuleType("ref_native_python")
    pkg.__path__ = ["/root/reference/openpose/native/python"]   # modules by path, no __init__ executed
    sys.modules["ref_native_python"] = pkg
    args = importlib.import_module("ref_native_python.args")
    utils = importlib.import_module("ref_native_python.utils")
    flags = []
    for a in args.get_parser()._actions:
        if a.dest == "help":
            continue
        flags.append({"flag": a.option_strings[0], "dest": a.dest, "default": a.default,
                      "type": getattr(a.type, "__name__", str(a.type))})
    truth = {}
    for v in ("yes", "TRUE", "t", "Y", "1", "no", "False", "f", "N", "0", "maybe", ""):
        try:
            truth[v] = utils.str2bool(v)
        except Exception as e:
            truth[v] = type(e).__name__
    spec = importlib.util.spec_from_file_location("ref_movenet_common", "/root/reference/movenet/common.py")
    mvc = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mvc)                                  # constants only (movenet/common.py)
    movenet = {"KEYPOINT_DICT": mvc.KEYPOINT_DICT, "MODEL_URL": mvc.MODEL_URL,
               "MIN_CROP_KEYPOINT_SCORE": mvc.MIN_CROP_KEYPOINT_SCORE,
               "KEYPOINT_EDGE_INDS_TO_COLOR": [[a, b, c] for (a, b), c in mvc.KEYPOINT_EDGE_INDS_TO_COLOR.items()]}
    meta = {"movenet_common": movenet, "flags": flags, "str2bool": truth,
            "get_color": [list(utils.get_color(i)) for i in range(12)],
            "get_color_fast": [list(utils.get_color_fast(i)) for i in range(12)]}
    with open(os.path.join(HERE, "callers_ref.json"), "w") as f:
        json.dump(meta, f, indent=1)
    rng = np.random.default_rng(811)
    out = {}
    # resize_tensor: (h, w, c) crops as create_detections makes them (tracking/utils_track.py:54-55) -> 25 x 50
    for i, (h, w, c, x, y) in enumerate(((37, 19, 3, 25, 50), (2, 2, 1, 25, 50), (120, 61, 5, 25, 50),
                                          (9, 33, 2, 7, 4), (50, 25, 1, 25, 50))):
        img = rng.normal(size=(h, w, c)).astype(np.float32)
        out[f"rt{i}_in"] = img
        out[f"rt{i}_xy"] = np.array([x, y])
        out[f"rt{i}_out"] = utils.resize_tensor(img, x, y)
    a = rng.normal(size=(5, 16))
    b = rng.normal(size=(7, 16))
    out["cos_a"], out["cos_b"] = a, b
    out["cos_d"] = utils.cosine_distance(a, b)
    an = a / np.linalg.norm(a, axis=1, keepdims=True)
    bn = b / np.linalg.norm(b, axis=1, keepdims=True)
    out["cos_d_normalized"] = utils.cosine_distance(an, bn, True)
    np.savez_compressed(os.path.join(HERE, "callers_ref.npz"), **out)


if __name__ == "__main__":
    tables()
    callers()
    v2()
    v1()
    real_images()
    print("golden fixtures written to", HERE)
