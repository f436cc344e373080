"""MoveNet steps around the model (BASELINE configs[3]): the reference's
``movenet/utils/image_sequence_crop.py`` / ``movenet/common.py`` interface -- same function names, argument
meaning and the ``crop_region`` dict -- evaluated on the MI355X through libposehip, plus batched forms (the
reference handles one stream at a time) and the MultiPose head decode BASELINE configs[3] names.

The model itself is a remote TF-Hub graph (``common.MODEL_URL``); ``run_inference`` therefore takes any
callable ``movenet(input_int32 [1,S,S,3]) -> {'output_0': [1,1,17,3]}``.  There is no CPU fallback.
"""
import ctypes as C
from typing import Callable, Sequence

import numpy as np

from . import _ffi
from .engine import Engine

# movenet/common.py:2-48 (data)
KEYPOINT_DICT = {
    'nose': 0, 'left_eye': 1, 'right_eye': 2, 'left_ear': 3, 'right_ear': 4, 'left_shoulder': 5,
    'right_shoulder': 6, 'left_elbow': 7, 'right_elbow': 8, 'left_wrist': 9, 'right_wrist': 10, 'left_hip': 11,
    'right_hip': 12, 'left_knee': 13, 'right_knee': 14, 'left_ankle': 15, 'right_ankle': 16}
KEYPOINT_EDGE_INDS_TO_COLOR = {
    (0, 1): 'm', (0, 2): 'c', (1, 3): 'm', (2, 4): 'c', (0, 5): 'm', (0, 6): 'c', (5, 7): 'm', (7, 9): 'm',
    (6, 8): 'c', (8, 10): 'c', (5, 6): 'y', (5, 11): 'm', (6, 12): 'c', (11, 12): 'y', (11, 13): 'm',
    (13, 15): 'm', (12, 14): 'c', (14, 16): 'c'}
MODEL_URL = "https://tfhub.dev/google/movenet/singlepose"
MIN_CROP_KEYPOINT_SCORE = 0.2

_KEYS = ('y_min', 'x_min', 'y_max', 'x_max', 'height', 'width')
_engine = None


def _eng() -> Engine:
    global _engine
    if _engine is None:
        _engine = Engine(0)
    return _engine


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def region_to_array(crop_region: dict) -> np.ndarray:
    return np.array([crop_region[k] for k in _KEYS], np.float64)


def array_to_region(row) -> dict:
    return {k: float(v) for k, v in zip(_KEYS, row)}


def init_crop_region(image_height, image_width) -> dict:
    """Default crop: the full image padded to a square (``image_sequence_crop.py:15-40``).  Host arithmetic --
    six divisions -- identical to what ``ph_movenet_update`` falls back to on the device."""
    if image_width > image_height:
        box_height, box_width = image_width / image_height, 1.0
        y_min, x_min = (image_height / 2 - image_width / 2) / image_height, 0.0
    else:
        box_height, box_width = 1.0, image_height / image_width
        y_min, x_min = 0.0, (image_width / 2 - image_height / 2) / image_width
    return {'y_min': y_min, 'x_min': x_min, 'y_max': y_min + box_height, 'x_max': x_min + box_width,
            'height': box_height, 'width': box_width}


def crop_and_resize_batch(frames: np.ndarray, crop_regions: Sequence[dict], crop_size) -> np.ndarray:
    """frames u8 [B,H,W,3] + one crop region per frame -> int32 [B,S,S,3] (``tf.image.crop_and_resize`` bilinear,
    extrapolation 0, then the int32 cast of ``run_inference``)."""
    frames = np.ascontiguousarray(frames, np.uint8)
    if frames.ndim != 4 or frames.shape[3] != 3:
        raise ValueError("frames must be u8 [B,H,W,3]")
    size = int(crop_size[0])
    if int(crop_size[1]) != size:
        raise ValueError("crop_size must be square (the MoveNet inputs are 192x192 / 256x256)")
    B, H, W, _ = frames.shape
    crop = np.stack([region_to_array(r) for r in crop_regions])
    out = np.empty((B, size, size, 3), np.int32)
    e = _eng()
    _ffi.check(e._lib.ph_movenet_crop_resize(e._ctx, _ptr(frames), B, H, W, _ptr(crop), size, _ptr(out), 0),
               "ph_movenet_crop_resize")
    return out


def crop_and_resize(image, crop_region: dict, crop_size) -> np.ndarray:
    """``image_sequence_crop.py:153-159`` for ``image`` [1,H,W,3] (as after ``tf.expand_dims``) or [H,W,3];
    returns int32 [1,S,S,3] (the reference casts right after the call, :171)."""
    image = np.asarray(image)
    if image.ndim == 3:
        image = image[None]
    return crop_and_resize_batch(image, [crop_region], crop_size)


def update_and_determine_batch(keypoints: np.ndarray, crop_regions: Sequence[dict], image_height, image_width):
    """keypoints f32 [B,1,17,3] (model outputs in crop coordinates) -> (keypoints in image coordinates
    [B,1,17,3], next crop region per stream): the update loop of ``run_inference`` (:178-186) followed by
    ``determine_crop_region`` (:94-150), one GPU thread per stream."""
    k = np.ascontiguousarray(keypoints, np.float32).reshape(-1, 17, 3).copy()
    B = k.shape[0]
    cin = np.stack([region_to_array(r) for r in crop_regions])
    cout = np.empty((B, 6), np.float64)
    e = _eng()
    _ffi.check(e._lib.ph_movenet_update(e._ctx, _ptr(k), _ptr(cin), B, int(image_height), int(image_width),
                                        _ptr(cout), 0), "ph_movenet_update")
    return k.reshape(B, 1, 17, 3), [array_to_region(r) for r in cout]


def determine_crop_region(keypoints, image_height, image_width) -> dict:
    """``image_sequence_crop.py:94-150`` for keypoints [1,1,17,3] already in image coordinates."""
    unit = {'y_min': 0.0, 'x_min': 0.0, 'y_max': 1.0, 'x_max': 1.0, 'height': 1.0, 'width': 1.0}
    return update_and_determine_batch(np.asarray(keypoints, np.float32).reshape(1, 1, 17, 3), [unit], image_height,
                                      image_width)[1][0]


def run_inference(movenet: Callable, image: np.ndarray, crop_region: dict, crop_size):
    """``image_sequence_crop.py:162-187``: crop, run the model, map its output back to image coordinates."""
    image_height, image_width, _ = image.shape
    input_image = crop_and_resize(image, crop_region, crop_size)
    outputs = movenet(input_image)
    k = np.asarray(outputs['output_0'], np.float32)
    return update_and_determine_batch(k.reshape(1, 1, 17, 3), [crop_region], image_height, image_width)[0][0:1].reshape(
        1, 1, 17, 3)


def multipose_decode(center, kpt_heat, kpt_regr, kpt_offs, box_size, max_people=6, min_center=0.0) -> np.ndarray:
    """MultiPose heads [B,G,G,...] -> [B,max_people,56] (17 x (y, x, score), box ymin xmin ymax xmax score).
    Not part of the reference tree; specification and status: ``oracle/movenet.py:multipose_decode`` /
    ``include/posehip.h``."""
    arrs = [np.ascontiguousarray(a, np.float32) for a in (center, kpt_heat, kpt_regr, kpt_offs, box_size)]
    B, G = arrs[0].shape[0], arrs[0].shape[1]
    want = [(B, G, G), (B, G, G, 17), (B, G, G, 34), (B, G, G, 34), (B, G, G, 2)]
    for a, w in zip(arrs, want):
        if a.shape != w:
            raise ValueError(f"head shape {a.shape}, expected {w}")
    out = np.empty((B, int(max_people), 56), np.float32)
    e = _eng()
    _ffi.check(e._lib.ph_movenet_multipose_decode(e._ctx, *[_ptr(a) for a in arrs], B, G, int(max_people),
                                                  C.c_float(min_center), _ptr(out), 0), "ph_movenet_multipose_decode")
    return out
