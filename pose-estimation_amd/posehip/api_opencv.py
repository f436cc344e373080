"""Drop-in for the reference's cv2.dnn estimator class ``PyOpenPoseOpenCV``
(``openpose/opencv/skeleton.py:20-227``): same constructor, ``load``,
``predict``, ``postprocess_single``, ``postprocess_multi``, ``draw_skeleton``,
no ``cv2`` import.  ``predict`` = blobFromImage + net.forward on the MI355X
(preprocess kernel + MFMA network); ``postprocess_multi`` = the V1 decode kernels
(csrc/decode_v1.hip) on the net output still resident in HBM.

Addition over the reference: ``predict_batch(frames)``; ``load(...,
random_weights=seed)`` when no model files exist.
"""
import os
from typing import List, Optional, Tuple

import numpy as np

from . import _ffi, caffe_io, draw, graph
from .engine import Engine
from .tables import COLORS, MAPIDX, get_body_parts_and_pose_pairs  # noqa: F401


class _Net:
    """What callers touch on ``self.net`` (``getPerfProfile``, skeleton.py:215)."""

    def __init__(self, engine):
        self.engine = engine
        self.last_ms = 0.0

    def getPerfProfile(self):
        return self.last_ms, []

    def empty(self):
        return self.engine.plan is None


class PyOpenPoseOpenCV:
    """Class to run openpose on the MI355X behind the reference's cv2 interface."""

    def __init__(self, image_width: int = 368, image_height: int = 368, body_parts: Optional[dict] = None,
                 pose_pairs: Optional[list] = None, conf_thres: float = 0.1) -> None:
        super().__init__()
        self.net = None
        self.image_width = image_width
        self.image_height = image_height
        self.original_image_width = None
        self.original_image_height = None
        self.body_parts = body_parts  # name: id
        self.pose_pairs = pose_pairs  # name : name
        self.conf_thres = conf_thres
        self._parsed = None
        self._weights = None
        self._batch = 0
        self._last_b = 0

    def load(self, proto_file: str, weights_file: str, cuda: bool = True, random_weights: Optional[int] = None,
             device: int = 0) -> None:
        """Loads the Caffe model (``readNetFromCaffe``, skeleton.py:39-53).  ``cuda``
        is accepted for compatibility; the network always runs on the MI355X."""
        if os.path.exists(proto_file):
            with open(proto_file) as f:
                self._parsed = caffe_io.parse_prototxt(f.read())
        elif random_weights is not None:
            n_heat = len(self.body_parts) if self.body_parts else 19
            gen = caffe_io.body25_prototxt if n_heat == 26 else caffe_io.coco_prototxt
            self._parsed = caffe_io.parse_prototxt(gen(self.image_height, self.image_width))
        else:
            raise FileNotFoundError(proto_file)
        if os.path.exists(weights_file):
            self._weights = caffe_io.read_caffemodel(weights_file)
        elif random_weights is not None:
            self._weights = graph.random_weights(self._parsed, seed=int(random_weights))
        else:
            raise FileNotFoundError(weights_file)
        self.net = _Net(Engine(device))
        self._batch = 0

    def _ensure(self, batch):
        if batch > self._batch:
            plan = graph.build_plan(self._parsed, self._weights, height=self.image_height, width=self.image_width)
            self.net.engine.build(plan, batch)
            self._batch = batch

    def predict(self, image: np.ndarray) -> np.ndarray:
        """u8 [H,W,3] BGR -> f32 [1,C,H/8,W/8] (skeleton.py:55-75)."""
        return self.predict_batch(image[None])

    def predict_batch(self, images: np.ndarray) -> np.ndarray:
        images = np.ascontiguousarray(images)
        self.original_image_width = images.shape[2]
        self.original_image_height = images.shape[1]
        B = images.shape[0]
        self._ensure(B)
        eng = self.net.engine
        eng.preprocess(images, _ffi.PH_PRE_CV2)
        ms = eng.forward_timed(B)
        self.net.last_ms = float(ms.sum())
        self._last_b = B
        return eng.net_out(B)

    def _resident(self, prediction):
        """Make sure the ctx holds `prediction` (it does right after predict())."""
        eng = self.net.engine if self.net is not None else None
        if eng is None:
            self.net = _Net(Engine(0))
            eng = self.net.engine
        eng.set_net_out(np.ascontiguousarray(prediction, np.float32))
        return eng

    def postprocess_single(self, prediction: np.ndarray) -> list:
        """Global arg-max per heat map (skeleton.py:77-100)."""
        assert(len(self.body_parts) <= prediction.shape[1])
        eng = self._resident(prediction[:1])
        vals, xs, ys = eng.argmax_maps(1, prediction.shape[1], prediction.shape[3])
        points = []
        for i in range(len(self.body_parts)):
            conf = float(vals[0, i])
            x = (self.original_image_width * int(xs[0, i])) / prediction.shape[3]
            y = (self.original_image_height * int(ys[0, i])) / prediction.shape[2]
            points.append((int(x), int(y)) if conf > self.conf_thres else None)
        return points

    def postprocess_multi(self, prediction: np.ndarray) -> Tuple[np.ndarray, list]:
        """Multi-person decode (skeleton.py:102-154) -> (keypoints_list f64 [K,3],
        personwise_keypoints: list of f64 [len(body_parts)])."""
        assert(len(self.body_parts) <= prediction.shape[1])
        eng = self._resident(prediction[:1])
        res = eng.decode_v1(self.body_parts, self.pose_pairs, MAPIDX, 1, self.original_image_width,
                            self.original_image_height, self.conf_thres)
        for k in res[0][2]["invalid_pairs"]:
            print("No Connection : k = {}".format(k))       # utils.py:127
        return res[0][0], res[0][1]

    def postprocess_multi_batch(self, prediction: np.ndarray) -> List[Tuple[np.ndarray, list]]:
        eng = self._resident(prediction)
        res = eng.decode_v1(self.body_parts, self.pose_pairs, MAPIDX, prediction.shape[0],
                            self.original_image_width, self.original_image_height, self.conf_thres)
        return [(r[0], r[1]) for r in res]

    def draw_skeleton(self, points: Optional[list] = None, personwise_keypoints: Optional[List[np.ndarray]] = None,
                      keypoints_list: Optional[np.ndarray] = None, image: Optional[np.ndarray] = None,
                      show_image: bool = False, output_image_path: Optional[str] = None):
        if image is None:
            return None
        if points is not None:
            for pair in self.pose_pairs:
                assert(pair[0] in self.body_parts)
                assert(pair[1] in self.body_parts)
                a, b = points[self.body_parts[pair[0]]], points[self.body_parts[pair[1]]]
                if a and b:
                    draw.line(image, a, b, (0, 255, 0), 3)
                    draw.disc(image, a, 3, (0, 0, 255))
                    draw.disc(image, b, 3, (0, 0, 255))
        elif personwise_keypoints is not None and keypoints_list is not None:
            for i in range(keypoints_list.shape[0]):
                draw.disc(image, keypoints_list[i][0:2], 5, (255, 105, 180))
            for idx, pose_pair in enumerate(self.pose_pairs):
                for person in personwise_keypoints:
                    index = person[np.array([self.body_parts[x] for x in pose_pair])]
                    if -1 in index:
                        continue
                    B_ = np.int32(keypoints_list[index.astype(int), 0])
                    A_ = np.int32(keypoints_list[index.astype(int), 1])
                    draw.line(image, (B_[0], A_[0]), (B_[1], A_[1]), COLORS[idx], 3)
        if output_image_path is not None:
            assert isinstance(output_image_path, str)
            from PIL import Image
            Image.fromarray(np.ascontiguousarray(image[:, :, ::-1])).save(output_image_path)
        return image
