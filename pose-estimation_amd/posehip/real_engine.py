"""Opportunistic check against the REAL engine (last row): where ``cv2`` is importable and
``$POSEHIP_MODEL_DIR`` holds the COCO Caffe model, run the reference's own sequence
(``openpose/opencv/skeleton.py:50,66-75``: readNetFromCaffe -> blobFromImage -> forward) on a few images and
report how far posehip's maps and keypoints are from it.  Never a dependency: ``run`` returns None when either
is missing.  Both nets' maps go through the SAME decode (posehip's ``postprocess_multi``), so the keypoint error
measures what the map difference does to the peaks, not a difference between decoders."""
import glob
import os

import numpy as np


def compare(ref_maps, got_maps, ref_kp, got_kp, match_px=8.0):
    """ref/got_maps: f32 [1,C,Hs,Ws]; ref/got_kp: keypoints_list [K,3] (x, y, score) in image pixels.
    -> {'heatmap_max_abs_err', 'keypoints_ref', 'keypoints_got', 'matched', 'keypoint_max_px_err'}: every reference
    keypoint is paired with its nearest posehip keypoint not yet taken, within `match_px`."""
    ref_kp, got_kp = np.asarray(ref_kp, np.float64).reshape(-1, 3), np.asarray(got_kp, np.float64).reshape(-1, 3)
    free, dists = np.ones(len(got_kp), bool), []
    for x, y, _ in ref_kp:
        if not free.any():
            break
        d = np.where(free, np.hypot(got_kp[:, 0] - x, got_kp[:, 1] - y), np.inf)
        j = int(np.argmin(d))
        if d[j] <= match_px:
            free[j] = False
            dists.append(float(d[j]))
    return {"heatmap_max_abs_err": float(np.abs(np.asarray(ref_maps) - np.asarray(got_maps)).max()),
            "keypoints_ref": len(ref_kp), "keypoints_got": len(got_kp), "matched": len(dists),
            "keypoint_max_px_err": max(dists) if dists else None}


def run(model_dir, image_glob, net_size=(368, 368), max_images=4):
    """-> list of `compare` dicts (one per image) or None when cv2 / the model / the images are not there."""
    proto = os.path.join(model_dir or "", "pose", "coco", "pose_deploy_linevec.prototxt")
    weights = os.path.join(model_dir or "", "pose", "coco", "pose_iter_440000.caffemodel")
    files = sorted(glob.glob(image_glob))[:max_images]
    try:
        import cv2
    except Exception:
        return None
    if not (files and os.path.exists(proto) and os.path.exists(weights)):
        return None
    from .api_opencv import PyOpenPoseOpenCV
    from .tables import get_body_parts_and_pose_pairs
    body_parts, pose_pairs = get_body_parts_and_pose_pairs("COCO")
    ours = PyOpenPoseOpenCV(net_size[0], net_size[1], body_parts, pose_pairs)
    ours.load(proto, weights)
    net = cv2.dnn.readNetFromCaffe(proto, weights)
    out = []
    for path in files:
        image = cv2.imread(path)
        net.setInput(cv2.dnn.blobFromImage(image, 1.0 / 255, net_size, (0, 0, 0), swapRB=False, crop=False))
        ref = net.forward()
        got = ours.predict(image)
        got_kp, _ = ours.postprocess_multi(got)
        ref_kp, _ = ours.postprocess_multi(ref)
        out.append(dict(compare(ref, got, ref_kp, got_kp), image=os.path.basename(path)))
    return out
