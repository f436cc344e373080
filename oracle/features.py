"""ORACLE (test infrastructure only -- never imported by the product path).

CPU restatement of the tracker appearance feature:

* ``export_heatmaps``     -- what ``datum.poseHeatMaps`` holds and ``pose_heatmaps`` turns to HWC
                             (openpose/native/python/skeleton.py:64-76): the selected channel windows of the net
                             output, bicubic-resized to the net input size (oracle/decode_v2.upsample_bicubic =
                             upstream resizeAndMerge) and mapped by OpenPose's ``--heatmaps_scale``
                             (upstream poseExtractorNet.cpp, restated from the published 1.7.0 source;
                             **parity unpinned**: pyopenpose is absent).
* ``resize_tensor``       -- openpose/native/python/utils.py:101-110, the reference's own double loop.
                             Pinned: tests/golden/callers_ref.npz holds the reference's outputs.
* ``detection_features``  -- the crop + resize of ``create_detections``, tracking/utils_track.py:45-57, with the
                             scales of tracking/track.py:180-181.  Plain NumPy in the reference (exact oracle).
                             ``box[k] * s_w`` is evaluated in float64 (NumPy < 2 promotion of float32 scalar x
                             Python float, as in oracle/post_v2.py).
"""
import numpy as np

from . import decode_v2


def scale_values(v, heat, mode):
    """--heatmaps_scale on resized maps, fp32: 0 -> [-1,1], 1 -> [0,1], 2 -> rounded [0,255], 3 -> as is.
    `heat`: True for body-part / background maps ([0,1] data), False for PAFs ([-1,1] data)."""
    v = np.asarray(v, np.float32)
    f = np.float32
    if mode == 3:
        return v
    if heat:
        t = np.clip(v, f(0), f(1))
        if mode == 0:
            return t * f(2) - f(1)
        if mode == 2:
            return (t * f(255) + f(0.5)).astype(np.int32).astype(np.float32)
        return t
    t = np.clip(v, f(-1), f(1))
    if mode == 1:
        return t * f(0.5) + f(0.5)
    if mode == 2:
        r = ((t * f(128.5) + f(128.5)) + f(0.5)).astype(np.int32).astype(np.float32)
        return np.clip(r, f(0), f(255))
    return t


def export_heatmaps(net_frame, windows, n_heat, mode, h_target, w_target):
    """net_frame f32 [C,hs,ws] -> f32 [sum nc, Ht, Wt] (CHW, as datum.poseHeatMaps)."""
    planes = []
    for c0, nc in windows:
        for c in range(c0, c0 + nc):
            up = decode_v2.upsample_bicubic(net_frame[c:c + 1], h_target, w_target)[0]
            planes.append(scale_values(up, c < n_heat, mode))
    return np.stack(planes)


def resize_tensor(image, x, y):
    h, w, c = image.shape
    x_scale = x / max((w - 1), 1)
    y_scale = y / max((h - 1), 1)
    new = np.zeros([y, x, c])
    for i in range(y - 1):
        for j in range(x - 1):
            new[i + 1, j + 1] = image[1 + int(i / y_scale), 1 + int(j / x_scale)]
    return new


def detection_features(boxes, heatmaps_hwc, image_size, size=(25, 50)):
    """boxes [M,4] (l,t,r,b), heatmaps [Ht,Wt,C], image_size (w,h) -> list of flattened f64 features; an entry
    is the exception instance where the reference's loop raises (degenerate crop)."""
    s_h = heatmaps_hwc.shape[0] / image_size[1]
    s_w = heatmaps_hwc.shape[1] / image_size[0]
    out = []
    for box in boxes:
        ll = int(np.floor(float(box[0]) * s_w))
        tt = int(np.floor(float(box[1]) * s_h))
        rr = int(np.ceil(float(box[2]) * s_w))
        bb = int(np.ceil(float(box[3]) * s_h))
        crop = heatmaps_hwc[tt:bb, ll:rr]
        try:
            out.append(resize_tensor(crop, size[0], size[1]).flatten())
        except IndexError as e:
            out.append(e)
    return out
