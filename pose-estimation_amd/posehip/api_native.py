"""Drop-in for the reference's pyopenpose estimator classes
(``openpose/native/python/skeleton.py``): ``PyOpenPoseNativeBase`` (:17-370),
``PyOpenPoseNative`` (:373-514), ``OpenPosePoseExtractor`` (:517-608) and
``TORSO_IDS`` (:11) -- same constructor arguments, methods, properties and
attributes, no ``pyopenpose`` / ``cv2`` import.  ``op.Datum`` / ``op.WrapperPython``
are replaced by :class:`Datum` / :class:`Wrapper`, whose ``emplaceAndPop`` runs
preprocess -> MFMA network -> V2 decode on the MI355X through libposehip.

Additions over the reference: ``predict_batch(frames)`` (the reference always
runs batch 1) and the ``posehip_*`` keys of the params dict:
  posehip_random_weights : int seed -- no model files exist offline; without it a
                           missing ``model_folder`` raises FileNotFoundError
  posehip_device         : GPU index (default 0)
  posehip_batch          : max frames per call (default 1)
"""
import argparse
import math
import os
from typing import Any, List, Optional, Tuple, Union

import numpy as np

from . import _ffi, caffe_io, draw, graph
from .engine import Engine
from .tables import TORSO_IDS, V2_MODELS  # noqa: F401  (TORSO_IDS re-exported)

_MODEL_FILES = {
    "BODY_25": ("pose/body_25/pose_deploy.prototxt", "pose/body_25/pose_iter_584000.caffemodel"),
    "COCO": ("pose/coco/pose_deploy_linevec.prototxt", "pose/coco/pose_iter_440000.caffemodel"),
}


def get_color(idx):
    """openpose/native/python/utils.py:113-116."""
    idx = idx * 3
    return ((37 * idx) % 255, (17 * idx) % 255, (29 * idx) % 255)


class _Point:
    def __init__(self, x, y):
        self.x, self.y = x, y


class Datum:
    """The fields of ``op.Datum`` the reference reads or writes."""

    def __init__(self):
        self.cvInputData = None
        self.cvOutputData = None
        self.poseKeypoints = None
        self.poseScores = None
        self.poseHeatMaps = None
        self.poseRawHeatMaps = []
        self.netInputSizes = []
        self.customInputNetData = []


def net_input_size(net_resolution: str, image_w: int, image_h: int) -> Tuple[int, int]:
    """OpenPose ``--net_resolution`` semantics: "WxH", multiples of 16; a side given as -1 follows the image aspect
    ratio and is ROUNDED to the nearest multiple of 16 in float32, as upstream ``ScaleAndSizeExtractor::extract``
    does: ``16 * positiveIntRound(1/16.f * h * W0 / (float)H0)`` (1920x1080 at "-1x368": 654.2 -> 656)."""
    w, h = (int(v) for v in net_resolution.lower().split("x"))
    if w <= 0 and h <= 0:
        raise ValueError(f"net_resolution {net_resolution!r}: width and height cannot both be -1")
    f32 = np.float32
    if w <= 0:
        w = 16 * int(f32(f32(f32(1) / f32(16)) * f32(h)) * f32(image_w) / f32(image_h) + f32(0.5))
    elif h <= 0:
        h = 16 * int(f32(f32(f32(1) / f32(16)) * f32(w)) * f32(image_h) / f32(image_w) + f32(0.5))
    if w % 16 or h % 16 or w < 16 or h < 16:
        raise ValueError(f"net_resolution {net_resolution!r} must be multiples of 16")
    return w, h


def scale_net_to_output(image_w: int, image_h: int, net_w: int, net_h: int) -> float:
    """Upstream ``mScaleNetToOutput`` (poseExtractorCaffe.cpp forwardPass): keypoints on the net-input-sized maps
    -> input-image pixels.  Not 1/s: OpenPose rounds the scaled image size to whole pixels first."""
    s = min((net_w - 1) / float(max(image_w - 1, 1)), (net_h - 1) / float(max(image_h - 1, 1)))
    nw, nh = int(s * image_w + 0.5), int(s * image_h + 0.5)
    return float(np.float32(min((image_w - 1) / float(max(nw - 1, 1)), (image_h - 1) / float(max(nh - 1, 1)))))


class Wrapper:
    """Stands in for ``op.WrapperPython`` (configure / start / stop /
    emplaceAndPop, ``native/python/skeleton.py:424-447``)."""

    def __init__(self):
        self.params = {}
        self._engine = None
        self._net = None
        self._weights = None
        self._built = None          # (net_w, net_h, batch)
        self._model = None
        self._started = False
        self.want_raw = False       # fill datum.poseRawHeatMaps (one D2H of the net output)
        self._last = None           # geometry of the last emplaceAndPop (for track_features)
        self._feat = None           # its per-person features, computed on first use

    def configure(self, params: dict) -> None:
        self.params = dict(params)

    def start(self) -> None:
        p = self.params
        name = p.get("model_pose", "BODY_25")
        if name not in V2_MODELS:
            raise ValueError(f"model_pose {name!r} not supported (have {sorted(V2_MODELS)})")
        self._model = V2_MODELS[name]
        folder = p.get("model_folder", "/usr/local/src/openpose/models/")
        proto, weights = (os.path.join(folder, f) for f in _MODEL_FILES[name])
        if os.path.exists(proto) and os.path.exists(weights):
            with open(proto) as f:
                self._net = caffe_io.parse_prototxt(f.read())
            self._weights = caffe_io.read_caffemodel(weights)
        elif p.get("posehip_random_weights") is not None:
            gen = caffe_io.body25_prototxt if name == "BODY_25" else caffe_io.coco_prototxt
            self._net = caffe_io.parse_prototxt(gen(368, 656))
            self._weights = graph.random_weights(self._net, seed=int(p["posehip_random_weights"]))
        else:
            raise FileNotFoundError(
                f"OpenPose model files not found ({proto}, {weights}); pass "
                "params['posehip_random_weights']=seed to run the same graph with random weights")
        if self._engine is None:
            self._engine = Engine(int(p.get("posehip_device", 0)))
        self._built = None
        self._started = True

    def stop(self) -> None:
        self._started = False

    def close(self) -> None:
        if self._engine is not None:
            self._engine.close()
            self._engine = None

    # ------------------------------------------------------------------
    def _ensure(self, net_w, net_h, batch):
        key = (net_w, net_h)
        if self._built is None or self._built[:2] != key or self._built[2] < batch:
            plan = graph.build_plan(self._net, self._weights, height=net_h, width=net_w)
            cap = max(batch, int(self.params.get("posehip_batch", 1)))
            self._engine.build(plan, cap)
            self._built = (net_w, net_h, cap)

    def _heatmap_windows(self):
        """Channel windows of the net output that datum.poseHeatMaps concatenates (parts, background, PAFs)."""
        p, model = self.params, self._model
        chans = []
        if p.get("heatmaps_add_parts"):
            chans.append((0, model.n_parts))
        if p.get("heatmaps_add_bkg"):
            chans.append((model.n_parts, 1))
        if p.get("heatmaps_add_PAFs"):
            chans.append((model.n_heat, 2 * model.n_pairs))
        return chans

    def track_features(self, frame: int, image_size=None, size=(25, 50)):
        """Addition: the tracker's appearance feature of every person of frame `frame` of the last
        emplaceAndPop -- what ``create_detections`` (tracking/utils_track.py:45-57) cuts out of ``pose_heatmaps``
        and resamples with ``resize_tensor`` -- computed on the GPU from the resident net output without
        materialising the upsampled maps.  -> (features f64 [M, y*x*C], status i32 [M])"""
        last = self._last
        if last is None:
            raise RuntimeError("no frame has been processed")
        chans = self._heatmap_windows()
        if not chans:
            raise ValueError("heat maps are not enabled (heatmaps_add_parts / _bkg / _PAFs)")
        image_size = tuple(image_size) if image_size is not None else last["image_size"]
        key = (image_size, tuple(size))
        if self._feat is None or self._feat[0] != key:
            feat, status = self._engine.pose_features(
                last["B"], last["max_people"], self._model.n_parts, TORSO_IDS, chans, self._model.n_heat,
                int(self.params.get("heatmaps_scale", 2)), last["net_h"], last["net_w"], image_size,
                sum(last["counts"]), size)
            self._feat = (key, feat, status)
        _, feat, status = self._feat
        lo = sum(last["counts"][:frame])
        hi = lo + last["counts"][frame]
        return feat[lo:hi].reshape(hi - lo, -1).astype(np.float64), status[lo:hi]

    def emplaceAndPop(self, datums: List[Datum]) -> bool:
        """Synchronous full pipeline on a list of datums (all frames must share
        one image size); results are written into each datum."""
        if not self._started:
            raise RuntimeError("OpenPose wrapper not started")
        datums = list(datums)
        if not datums:
            return False
        custom = datums[0].customInputNetData
        model, eng, p = self._model, self._engine, self.params
        frames = [np.ascontiguousarray(d.cvInputData) for d in datums]
        H0, W0 = frames[0].shape[:2]
        B = len(frames)
        if custom is not None and len(custom) > 0:
            # decode-only: injected net output [1,C,hs,ws] per datum (predict_from_hm)
            hm = np.concatenate([np.asarray(d.customInputNetData[0], np.float32) for d in datums], axis=0)
            if hm.ndim != 4 or hm.shape[1] < model.n_channels:
                raise ValueError(f"custom net data must be [1,{model.n_channels},hs,ws], got {hm.shape}")
            eng.set_net_out(hm)
            net_h, net_w = hm.shape[2] * 8, hm.shape[3] * 8
            raw = hm
        else:
            for f in frames:
                if f.dtype != np.uint8 or f.ndim != 3 or f.shape != (H0, W0, 3):
                    raise ValueError("cvInputData must be u8 [H,W,3] BGR frames of one size")
            net_w, net_h = net_input_size(p.get("net_resolution", "-1x368"), W0, H0)
            self._ensure(net_w, net_h, B)
            eng.preprocess(np.stack(frames), _ffi.PH_PRE_NATIVE)
            eng.forward(B)
            raw = eng.net_out(B) if self.want_raw else None
        scale = scale_net_to_output(W0, H0, net_w, net_h)
        res, _ = eng.decode_v2(model, B, net_h, net_w, scale=scale,
                               max_people=int(p.get("number_people_max", 0)) or 128)
        want_hm = p.get("heatmaps_add_parts") or p.get("heatmaps_add_bkg") or p.get("heatmaps_add_PAFs")
        chans = self._heatmap_windows()
        hm_mode = int(p.get("heatmaps_scale", 2))     # OpenPose's flag default is 2; the reference passes 1
        max_people = int(p.get("number_people_max", 0)) or 128
        self._last = dict(B=B, net_h=net_h, net_w=net_w, max_people=max_people, image_size=(W0, H0),
                          counts=[len(res[b][1]) for b in range(B)])
        self._feat = None
        for b, d in enumerate(datums):
            kp, sc = res[b]
            d.poseKeypoints = kp if len(sc) else None
            d.poseScores = sc if len(sc) else None
            d.netInputSizes = [_Point(net_w, net_h)]
            d.poseRawHeatMaps = [raw[b:b + 1]] if raw is not None else []
            d._ph_frame = b
            if want_hm:
                d.poseHeatMaps = np.concatenate([eng.upsample_maps(b, c0, nc, net_h, net_w, model.n_heat, hm_mode)
                                                 for c0, nc in chans])
            else:
                d.poseHeatMaps = None
            img = np.zeros((H0, W0, 3), np.uint8) if p.get("disable_blending") else frames[b].copy()
            if p.get("render_pose", 0) != 0 and d.poseKeypoints is not None:
                draw.render_pose(img, d.poseKeypoints, model.pairs)
            d.cvOutputData = img
        return True


class PyOpenPoseNativeBase:
    """Properties and utilities (``native/python/skeleton.py:17-370``)."""

    def __init__(self, skel_thres: float = 0.0, max_true_body: int = 2, patch_offset: int = 2) -> None:
        super().__init__()
        self.skel_thres = skel_thres
        self.max_true_body = max_true_body
        self.patch_offset = patch_offset
        self.datum = Datum()
        self._pose_empty = False
        self._pose_heatmaps = None
        self._pose_keypoints_3d = None
        self._pose_bounding_box = None
        self._pose_bounding_box_int = None
        self.reset()

    def reset(self) -> None:
        self._pose_empty = False
        self._pose_heatmaps = None
        self._pose_keypoints_3d = None
        self._pose_bounding_box = None
        self._pose_bounding_box_int = None
        return

    @property
    def opencv_image(self) -> np.ndarray:
        return self.datum.cvOutputData

    @property
    def pose_empty(self):
        return self._pose_empty

    @property
    def pose_scores(self) -> Optional[np.ndarray]:
        return self.datum.poseScores

    @property
    def pose_heatmaps(self) -> Optional[np.ndarray]:
        # [op_h, op_w, C]; zeros [h, w, J] when heat maps were not requested (:64-76)
        if self._pose_heatmaps is None:
            if self.datum.poseHeatMaps is None:
                j = self.datum.poseKeypoints.shape[1]
                w = self.datum.netInputSizes[0].x
                h = self.datum.netInputSizes[0].y
                self._pose_heatmaps = np.zeros((h, w, j))
            else:
                self._pose_heatmaps = np.moveaxis(self.datum.poseHeatMaps, 0, -1)
        return self._pose_heatmaps

    @property
    def pose_keypoints(self) -> Optional[np.ndarray]:
        return self.datum.poseKeypoints

    @property
    def pose_keypoints_3d(self) -> Optional[np.ndarray]:
        return self._pose_keypoints_3d

    @property
    def pose_bounding_box(self) -> Optional[np.ndarray]:
        # torso box over TORSO_IDS joints with score != 0 (:89-113)
        if self._pose_empty:
            return None
        if self._pose_bounding_box is None:
            bb = []
            for pose_keypoints in self.pose_keypoints:
                u = pose_keypoints[TORSO_IDS, 0]
                v = pose_keypoints[TORSO_IDS, 1]
                s = pose_keypoints[TORSO_IDS, 2]
                if (s != 0).any():
                    u_min = u[s != 0].min()
                    u_max = u[s != 0].max()
                    v_min = v[s != 0].min()
                    v_max = v[s != 0].max()
                    if u_min == u_max:
                        u_max += 1
                    if v_min == v_max:
                        v_max += 1
                else:
                    u_min = v_min = 0
                    u_max = v_max = 1
                bb.append(np.asarray([u_min, v_min, u_max, v_max]))
            self._pose_bounding_box = np.stack(bb)
        return self._pose_bounding_box

    @property
    def pose_bounding_box_int(self) -> Optional[np.ndarray]:
        if self._pose_empty:
            return None
        if self._pose_bounding_box_int is None:
            _bb = self.pose_bounding_box
            if _bb is not None:
                u_min = np.floor(_bb[:, 0]).astype(int)
                v_min = np.floor(_bb[:, 1]).astype(int)
                u_max = np.ceil(_bb[:, 2]).astype(int)
                v_max = np.ceil(_bb[:, 3]).astype(int)
                self._pose_bounding_box_int = np.stack([u_min, v_min, u_max, v_max], axis=1)
        return self._pose_bounding_box_int

    @staticmethod
    def get_3d_skeleton(skeleton: np.ndarray, depth_img: np.ndarray, intr_mat: Union[list, np.ndarray],
                        depth_scale: float = 1e-3, patch_offset: int = 2) -> np.ndarray:
        # depth-patch mean + pinhole back-projection (:130-160)
        if isinstance(intr_mat, list):
            fx, fy, cx, cy = intr_mat[0], intr_mat[4], intr_mat[2], intr_mat[5]
        elif isinstance(intr_mat, np.ndarray):
            fx, fy, cx, cy = intr_mat[0, 0], intr_mat[1, 1], intr_mat[0, 2], intr_mat[1, 2]
        else:
            raise ValueError("Unknown intr_mat format.")
        H, W = depth_img.shape
        joints3d = []
        for x, y, _ in skeleton:
            # float64 from here on (what the reference computes under NumPy < 2 promotion; also what
            # Engine.pose_post / ph_pose_post computes on the device)
            x, y = float(x), float(y)
            patch = depth_img[max(0, int(y - patch_offset)):min(H, int(y + patch_offset)),
                              max(0, int(x - patch_offset)):min(W, int(x + patch_offset))]
            depth_avg = float(np.nanmean(patch)) if patch.size else float("nan")
            x3d = (x - cx) / fx * depth_avg
            y3d = (y - cy) / fy * depth_avg
            joints3d.append([x3d * depth_scale, y3d * depth_scale, depth_avg * depth_scale])
        return np.array(joints3d)

    def convert_to_3d(self, depth_image: np.ndarray, intr_mat: Union[list, np.ndarray],
                      depth_scale: float = 1e-3) -> None:
        if not self.pose_empty:
            pose_keypoints_3d = []
            for pose_keypoint in self.pose_keypoints:
                pose_keypoints_3d.append(self.get_3d_skeleton(
                    skeleton=pose_keypoint, depth_img=depth_image, intr_mat=intr_mat,
                    depth_scale=depth_scale, patch_offset=self.patch_offset))
            self._pose_keypoints_3d = np.asarray(pose_keypoints_3d)
        return

    @staticmethod
    def save_2d_skeleton(keypoints: Optional[np.ndarray], scores: Optional[np.ndarray], save_path: str) -> None:
        # one CSV row per person: score, x0,y0,s0, ... (:186-199); empty file when nobody
        os.makedirs(os.path.dirname(save_path), exist_ok=True)
        if keypoints is None:
            open(save_path, 'a').close()
        else:
            M, _, _ = keypoints.shape
            data = np.concatenate([scores.reshape((M, 1)), keypoints.reshape((M, -1))], axis=1)
            np.savetxt(save_path, data, delimiter=',')

    def save_pose_keypoints(self, save_path: str) -> None:
        self.save_2d_skeleton(keypoints=self.pose_keypoints, scores=self.pose_scores, save_path=save_path)
        return

    def save_3d_pose_keypoints(self, save_path: str) -> None:
        self.save_2d_skeleton(keypoints=self.pose_keypoints_3d, scores=self.pose_scores, save_path=save_path)
        return

    @staticmethod
    def _draw_text_on_skeleton_image(image: Optional[np.ndarray] = None,
                                     scores: Optional[np.ndarray] = None) -> np.ndarray:
        return image  # text rendering needs cv2.putText; omitted

    @staticmethod
    def _draw_depth_on_skeleton_image(image: Optional[np.ndarray] = None,
                                      depth: Optional[np.ndarray] = None) -> np.ndarray:
        if depth is not None and image is not None:
            g = np.clip(np.abs(depth.astype(np.float32)) * 0.065, 0, 255).astype(np.uint8)
            colormap = np.stack([g // 3, g // 2, g], axis=-1)   # stand-in for COLORMAP_INFERNO
            image = np.clip(image.astype(np.float32) * 0.7 + colormap.astype(np.float32) * 0.7, 0, 255).astype(np.uint8)
        return image

    @staticmethod
    def _draw_bounding_box_on_skeleton_image(image: np.ndarray, boxes: Optional[np.ndarray]) -> np.ndarray:
        if boxes is not None:
            for bb in boxes:
                image = draw.rectangle(image, bb[0:2], bb[2:4], (0, 125, 0), 2)
        return image

    @staticmethod
    def _draw_tracking_bounding_box_image(image: np.ndarray, tracks: Optional[list] = None) -> np.ndarray:
        if tracks is not None:
            for track in tracks:
                if track.is_activated:
                    try:
                        bb = track.to_tlbr()
                    except AttributeError:
                        bb = track.tlbr
                    try:
                        l, t, r, b = bb
                        x1, y1 = int(np.floor(l)), int(np.floor(t))
                        x2, y2 = int(np.ceil(r)), int(np.ceil(b))
                        sub = image[y1:y2, x1:x2]
                        rect = np.ones(sub.shape, dtype=np.float32) * np.array((60, 130, 0), dtype=np.float32)
                        image[y1:y2, x1:x2] = np.clip(sub * 0.5 + rect * 0.5 + 1.0, 0, 255).astype(np.uint8)
                    except TypeError:
                        pass
        return image

    def display(self, win_name: str, speed: int = 1000, scale: float = 1.0,
                depth_image: Optional[np.ndarray] = None, bounding_box: bool = False,
                tracks: Optional[list] = None, ori_image: Optional[np.ndarray] = None
                ) -> Tuple[bool, Optional[np.ndarray]]:
        image = self.opencv_image
        image = self._draw_depth_on_skeleton_image(image, depth_image)
        if bounding_box:
            image = self._draw_bounding_box_on_skeleton_image(image, self.pose_bounding_box_int)
        if tracks is not None:
            image = self._draw_tracking_bounding_box_image(image, tracks)
        image = draw.resize_nearest(image, int(image.shape[1] * scale), int(image.shape[0] * scale))
        if ori_image is not None:
            _image = draw.resize_nearest(ori_image, int(ori_image.shape[1] * scale), int(ori_image.shape[0] * scale))
            image = np.concatenate([_image, image], axis=0)
        return True, image

    def save_skeleton_image(self, save_path: str, depth: Optional[np.ndarray] = None) -> None:
        image = self.opencv_image
        image = self._draw_depth_on_skeleton_image(image, depth)
        os.makedirs(os.path.dirname(save_path), exist_ok=True)
        from PIL import Image
        Image.fromarray(np.ascontiguousarray(image[:, :, ::-1])).save(save_path)
        return


class PyOpenPoseNative(PyOpenPoseNativeBase):
    """``native/python/skeleton.py:373-514``."""

    def __init__(self, params: Optional[dict] = None, skel_thres: float = 0.0, max_true_body: int = 2,
                 patch_offset: int = 2) -> None:
        super().__init__(skel_thres, max_true_body, patch_offset)
        if params is None:
            params = dict()
            params["model_folder"] = "/usr/local/src/openpose/models/"
            params["model_pose"] = "BODY_25"
            params["net_resolution"] = "-1x368"
        params["disable_blending"] = True
        params["scale_number"] = 1
        params["body"] = 1
        self.params = params.copy()
        self.params_cout = params.copy()
        self.params_cin = params.copy()
        self.opWrapper = Wrapper()
        self.opWrapper.configure(self.params)
        self.opWrapper.start()
        self.filtered_skel = "0"
        self.reset()

    def reset(self) -> None:
        super().reset()
        self.filtered_skel = "0"
        return

    def configure(self, params: dict = None) -> None:
        if params is not None:
            self.opWrapper.stop()
            self.opWrapper.configure(params)
            self.opWrapper.start()
        return

    def predict(self, image: np.ndarray) -> None:
        self.reset()
        self.datum.cvInputData = image
        self.opWrapper.emplaceAndPop([self.datum])
        if self.datum.poseScores is None:
            self._pose_empty = True
        return

    def predict_batch(self, images) -> List[Tuple[Optional[np.ndarray], Optional[np.ndarray]]]:
        """Addition: one call for a batch of equally sized frames.  Returns
        [(poseKeypoints [M,J,3] | None, poseScores [M] | None)] per frame; the
        instance's properties then describe the LAST frame."""
        datums = []
        for img in images:
            d = Datum()
            d.cvInputData = img
            datums.append(d)
        self.reset()
        self.opWrapper.emplaceAndPop(datums)
        self.last_datums = datums
        self.adopt(datums[-1])
        return [(d.poseKeypoints, d.poseScores) for d in datums]

    def adopt(self, datum: Datum) -> None:
        """Addition: make one datum of the last ``predict_batch`` the instance's current frame, so
        the properties, ``filter_prediction`` and the writers act on it as after ``predict``."""
        self.reset()
        self.datum = datum
        if datum.poseScores is None:
            self._pose_empty = True

    def predict_hm(self, image: np.ndarray, hm_save_path: str) -> None:
        # raw net output -> .float file: f32 [ndim, d0..d3, data...] (:452-467)
        self.reset()
        self.datum.cvInputData = image
        self.opWrapper.want_raw = True
        try:
            self.opWrapper.emplaceAndPop([self.datum])
        finally:
            self.opWrapper.want_raw = False
        os.makedirs(os.path.dirname(hm_save_path), exist_ok=True)
        raw = self.datum.poseRawHeatMaps[0]
        np.insert(raw.reshape(-1), [0, 0, 0, 0, 0], [raw.ndim] + list(raw.shape)).tofile(hm_save_path)
        if self.datum.poseScores is None:
            self._pose_empty = True
        return

    def predict_from_hm(self, image: np.ndarray, heatmap: Optional[List[np.ndarray]] = None) -> None:
        self.reset()
        self.datum.cvInputData = image
        self.datum.customInputNetData = heatmap
        self.opWrapper.emplaceAndPop([self.datum])
        self.datum.customInputNetData = []
        if self.datum.poseScores is None:
            self._pose_empty = True
        return

    def filter_prediction(self) -> None:
        # keeps the reference's behaviour: only `filtered_skel` changes (:483-514)
        if self.pose_empty:
            pass
        else:
            scores = self.pose_scores
            scores_filtered = []
            keypoints_filtered = []
            max_score_idxs = np.argsort(scores)[-self.max_true_body:]
            for max_score_idx in max_score_idxs:
                if scores[max_score_idx] < self.skel_thres:
                    continue
                else:
                    keypoints_filtered.append(self.pose_keypoints[max_score_idx])
                    scores_filtered.append(scores[max_score_idx])
            if len(scores_filtered) == 0:
                self.reset()
            else:
                self._pose_keypoints = np.stack(keypoints_filtered, axis=0)
                self._pose_scores = np.stack(scores_filtered, axis=0)
            self.filtered_skel = f"{len(keypoints_filtered)}/{len(scores)}"
        return

    def pose_track_features(self, image_size: Optional[Tuple[int, int]] = None,
                            size: Tuple[int, int] = (25, 50)) -> Optional[np.ndarray]:
        """Addition: ``[M, size_y*size_x*C]`` float64, row m = the ``feature`` that
        ``create_detections(keypoints, scores, pose_bounding_box, pose_heatmaps, [s_w, s_h])``
        (tracking/utils_track.py:32-59, scales from tracking/track.py:180-181) builds for person m of the
        current frame, computed on the GPU.  ``image_size`` = (width, height) as the tracker receives it
        (default: the frame's own size).  Raises IndexError where the reference's ``resize_tensor`` does
        (a crop with fewer than two rows or columns)."""
        if self.pose_empty or self.datum.poseScores is None:
            return None
        feat, status = self.opWrapper.track_features(getattr(self.datum, "_ph_frame", 0), image_size, size)
        if (status < 0).any():
            raise IndexError("index 1 is out of bounds: a person's heat-map crop has fewer than 2 rows or columns")
        return feat


def read_heatmap_file(path: str) -> List[np.ndarray]:
    """Reader of the ``.float`` files ``predict_hm`` writes
    (``datacollection/utils.py:122-124``) -> ``[f32 [d0,d1,d2,d3]]``."""
    hm = np.fromfile(path, np.float32)
    return [hm[5:].reshape(int(hm[1]), int(hm[2]), int(hm[3]), int(hm[4]))]


class OpenPosePoseExtractor:
    """``native/python/skeleton.py:517-608``."""

    def __init__(self, args: argparse.Namespace) -> None:
        params = dict(
            model_folder=args.op_model_folder,
            model_pose=args.op_model_pose,
            net_resolution=args.op_net_resolution,
            heatmaps_add_parts=args.op_heatmaps_add_parts,
            heatmaps_add_bkg=args.op_heatmaps_add_bkg,
            heatmaps_add_PAFs=args.op_heatmaps_add_PAFs,
            heatmaps_scale=args.op_heatmaps_scale,
            render_pose=-1 if args.op_display > 0 else 0,
        )
        for k in ("posehip_random_weights", "posehip_device", "posehip_batch"):
            if getattr(args, k, None) is not None:
                params[k] = getattr(args, k)
        self.pyop = PyOpenPoseNative(params, args.op_skel_thres, args.op_max_true_body, args.op_patch_offset)

    def predict(self, image: np.ndarray, depth: np.ndarray, kpt_save_path: Optional[str] = None,
                skel_image_save_path: Optional[str] = None) -> None:
        self.pyop.predict(image)
        self.pyop.filter_prediction()
        if kpt_save_path is not None:
            self.pyop.save_pose_keypoints(kpt_save_path)
        if skel_image_save_path is not None:
            self.pyop.save_skeleton_image(skel_image_save_path, depth)

    def predict_hm(self, image: np.ndarray, hm_save_path: str) -> None:
        self.pyop.predict_hm(image, hm_save_path)

    def predict_from_hm(self, image: np.ndarray, heatmap: Optional[List[np.ndarray]] = None,
                        kpt_save_path: Optional[str] = None) -> None:
        self.pyop.predict_from_hm(image, heatmap)
        self.pyop.filter_prediction()
        if kpt_save_path is not None:
            self.pyop.save_pose_keypoints(kpt_save_path)

    def predict_3d(self, image: np.ndarray, depth: np.ndarray, intr_mat: np.ndarray, depth_scale: float = 1e-3,
                   kpt_save_path: Optional[str] = None, kpt_3d_save_path: Optional[str] = None,
                   skel_image_save_path: Optional[str] = None) -> None:
        self.predict(image, depth, kpt_save_path, skel_image_save_path)
        self.pyop.convert_to_3d(depth_image=depth, intr_mat=intr_mat, depth_scale=depth_scale)
        if kpt_3d_save_path is not None:
            self.pyop.save_3d_pose_keypoints(kpt_3d_save_path)

    def display(self, win_name: str = "1", speed: int = 1000, scale: int = 1.0,
                image: Optional[np.ndarray] = None, bounding_box: bool = False,
                tracks: list = None) -> Tuple[bool, Optional[np.ndarray]]:
        if scale > 0:
            if self.pyop.datum.poseScores is None:
                img = self.pyop.opencv_image
                if image is not None:
                    _image = draw.resize_nearest(image, int(image.shape[1] * scale), int(image.shape[0] * scale))
                    img = np.concatenate([_image, img], axis=0)
                return True, img
            else:
                return self.pyop.display(win_name, scale=scale, speed=speed, bounding_box=bounding_box,
                                         tracks=tracks, ori_image=image)
        else:
            return True, None
