"""The estimator-side half of the reference's tracker glue (``tracking/utils_track.py``):
``PoseDetection`` and ``create_detections``, the consumer of ``pose_heatmaps``.  The trackers themselves
(DeepSORT / StrongSORT / ByteTrack / OC-SORT submodules) are outside the path and not part of this repository;
``PoseDetection`` therefore does not derive from StrongSORT's ``Detection`` -- it carries the same attributes.
"""
from typing import Tuple

import numpy as np

from .utils import resize_tensor


class PoseDetection:
    """``tracking/utils_track.py:10-28``: pose (keypoints), tlwh (box from keypoints), confidence (pose score),
    feature (heat-map crop), feature_size."""

    def __init__(self, pose: np.ndarray, tlwh: np.ndarray, confidence: float, feature: np.ndarray) -> None:
        self.pose = pose
        self.tlwh = tlwh
        self.confidence = confidence
        self.feature = feature
        self.feature_size = feature.shape

    def to_tlbr(self):
        ret = np.asarray(self.tlwh, dtype=float).copy()
        ret[2:] += ret[:2]
        return ret


def _tlwh(box):
    return np.asarray([box[0], box[1], box[2] - box[0], box[3] - box[1]])


def create_detections(keypoints: np.ndarray, scores: np.ndarray, bounding_boxes: np.ndarray, heatmaps: np.ndarray,
                      heatmaps_scale: Tuple[float, float]) -> list:
    """Host form, as the reference (``tracking/utils_track.py:32-59``): per person, the crop of the HWC heat maps
    under the box (scaled to heat-map pixels), resampled to 25 x 50 by ``resize_tensor`` and flattened."""
    assert keypoints.shape[0] == scores.shape[0]
    assert keypoints.shape[0] == bounding_boxes.shape[0]
    s_w, s_h = heatmaps_scale
    detections = []
    for kpt, score, box in zip(keypoints, scores, bounding_boxes):
        ll, tt = int(np.floor(float(box[0]) * s_w)), int(np.floor(float(box[1]) * s_h))
        rr, bb = int(np.ceil(float(box[2]) * s_w)), int(np.ceil(float(box[3]) * s_h))
        feature = resize_tensor(heatmaps[tt:bb, ll:rr], 25, 50).flatten()
        detections.append(PoseDetection(kpt, _tlwh(box), score, feature))
    return detections


def create_detections_device(pyop, image_size: Tuple[int, int]) -> list:
    """Addition: the same detections for the estimator's current frame with the features computed on the GPU
    (``PyOpenPoseNative.pose_track_features``): the upsampled heat maps are neither materialised nor copied to the
    host.  ``image_size`` = (width, height) as ``Tracker.update`` receives it (``tracking/track.py:152-183``)."""
    if pyop.pose_empty:
        return []
    feats = pyop.pose_track_features(image_size)
    boxes = pyop.pose_bounding_box
    return [PoseDetection(k, _tlwh(b), s, f)
            for k, s, b, f in zip(pyop.pose_keypoints, pyop.pose_scores, boxes, feats)]
