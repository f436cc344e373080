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


# ----------------------------------------------------------------------------------------------------------------
# Batched NumPy forms of the reference's per-person helpers.  Same results as the loops at
# native/python/skeleton.py:89-160 (the device versions are ph_pose_post / Engine.pose_post).
# ----------------------------------------------------------------------------------------------------------------
def torso_boxes(keypoints: np.ndarray, torso_ids=TORSO_IDS) -> np.ndarray:
    """[M, J, 3] keypoints -> [M, 4] (u_min, v_min, u_max, v_max) over the torso joints whose score is not 0;
    a side of zero length grows by 1, a person without any torso joint gets (0, 0, 1, 1).  The result dtype
    follows what the reference's list-of-arrays ``np.stack`` produces: the keypoint dtype when every person has a
    box, int64 when nobody has, float64 when the two kinds mix."""
    t = np.asarray(keypoints)[:, list(torso_ids), :]
    seen = t[..., 2] != 0
    have = seen.any(axis=1)
    big = np.finfo(t.dtype).max if t.dtype.kind == "f" else np.iinfo(t.dtype).max
    uv = t[..., :2]
    lo = np.where(seen[..., None], uv, big).min(axis=1)
    hi = np.where(seen[..., None], uv, -big).max(axis=1)
    hi = np.where(hi == lo, hi + uv.dtype.type(1), hi)
    box = np.concatenate([lo, hi], axis=1)
    if have.all():
        return box
    out = np.empty(box.shape, np.int64 if not have.any() else np.float64)
    out[have] = box[have]
    out[~have] = (0, 0, 1, 1)
    return out


def integer_boxes(boxes: np.ndarray) -> np.ndarray:
    """Outward-rounded boxes: floor of the minima, ceil of the maxima, as Python ints (int64)."""
    b = np.asarray(boxes)
    return np.concatenate([np.floor(b[:, :2]), np.ceil(b[:, 2:])], axis=1).astype(int)


def _intrinsics(intr_mat) -> Tuple[float, float, float, float]:
    """(fx, fy, cx, cy) from a flat 3x3 list or a 3x3 array."""
    if isinstance(intr_mat, list):
        return intr_mat[0], intr_mat[4], intr_mat[2], intr_mat[5]
    if isinstance(intr_mat, np.ndarray):
        return intr_mat[0, 0], intr_mat[1, 1], intr_mat[0, 2], intr_mat[1, 2]
    raise ValueError("Unknown intr_mat format.")


def lift_to_3d(joints: np.ndarray, depth_img: np.ndarray, intr_mat, depth_scale: float = 1e-3,
               patch_offset: int = 2) -> np.ndarray:
    """[..., 3] (x, y, score) joints -> [..., 3] float64 camera-space points: mean depth of the patch
    [y - p, y + p) x [x - p, x + p) (clipped to the image, NaNs skipped, NaN when nothing is left) back-projected
    through the pinhole model and multiplied by depth_scale.  All joints at once: the patches are gathered with one
    fancy index and reduced with a validity mask."""
    fx, fy, cx, cy = _intrinsics(intr_mat)
    H, W = depth_img.shape
    xy = np.asarray(joints, dtype=np.float64)[..., :2].reshape(-1, 2)
    p = int(patch_offset)
    first = np.trunc(xy - p).astype(np.int64)                 # int(): towards zero
    last = np.trunc(xy + p).astype(np.int64)
    x0, y0 = np.maximum(first[:, 0], 0), np.maximum(first[:, 1], 0)
    x1, y1 = np.minimum(last[:, 0], W), np.minimum(last[:, 1], H)
    side = max(int(np.max(np.maximum(x1 - x0, y1 - y0), initial=0)), 1)
    step = np.arange(side)
    rows = y0[:, None] + step                                 # [N, side]
    cols = x0[:, None] + step
    inside = ((rows < y1[:, None]) & (rows >= 0))[:, :, None] & ((cols < x1[:, None]) & (cols >= 0))[:, None, :]
    patch = depth_img[np.clip(rows, 0, H - 1)[:, :, None], np.clip(cols, 0, W - 1)[:, None, :]].astype(np.float64)
    use = inside & ~np.isnan(patch)
    n = use.sum(axis=(1, 2))
    with np.errstate(invalid="ignore", divide="ignore"):
        z = np.where(use, patch, 0.0).sum(axis=(1, 2)) / n    # 0 / 0 = NaN: empty or all-NaN patch
    out = np.stack([(xy[:, 0] - cx) / fx * z, (xy[:, 1] - cy) / fy * z, z], axis=1) * depth_scale
    return out.reshape(np.asarray(joints).shape[:-1] + (3,))


def write_people_csv(path: str, per_person: Optional[np.ndarray], scores: Optional[np.ndarray]) -> None:
    """One text row per person: score, then the person's values flattened (``np.savetxt`` defaults, ',').  No
    people: the file is created empty (an existing one is left as it is)."""
    os.makedirs(os.path.dirname(path), exist_ok=True)
    if per_person is None:
        with open(path, 'a'):
            return
    people = len(per_person)
    np.savetxt(path, np.column_stack([np.reshape(scores, (people, 1)), np.reshape(per_person, (people, -1))]),
               delimiter=',')


def strongest_people(scores: np.ndarray, limit: int, floor: float) -> np.ndarray:
    """Indices of the `limit` highest scores in ascending score order, without those below `floor`."""
    order = np.argsort(scores)[-limit:]
    return order[np.asarray(scores)[order] >= floor]


class PyOpenPoseNativeBase:
    """Result properties and utilities of the native estimator (interface of
    ``native/python/skeleton.py:17-370``; the bodies are this package's own batched forms above)."""

    _LAZY = ("_pose_heatmaps", "_pose_keypoints_3d", "_pose_bounding_box", "_pose_bounding_box_int")

    def __init__(self, skel_thres: float = 0.0, max_true_body: int = 2, patch_offset: int = 2) -> None:
        super().__init__()
        self.skel_thres, self.max_true_body, self.patch_offset = skel_thres, max_true_body, patch_offset
        self.datum = Datum()
        PyOpenPoseNativeBase.reset(self)

    def reset(self) -> None:
        """Forget everything derived from the last frame."""
        self._pose_empty = False
        for name in self._LAZY:
            setattr(self, name, None)

    # ---- straight views of the datum
    opencv_image = property(lambda self: self.datum.cvOutputData)
    pose_empty = property(lambda self: self._pose_empty)
    pose_scores = property(lambda self: self.datum.poseScores)
    pose_keypoints = property(lambda self: self.datum.poseKeypoints)
    pose_keypoints_3d = property(lambda self: self._pose_keypoints_3d)

    # ---- derived, computed once per frame
    @property
    def pose_heatmaps(self) -> Optional[np.ndarray]:
        """Exported heat maps, channels last [h, w, C]; zeros [net_h, net_w, J] when none were requested."""
        if self._pose_heatmaps is None:
            d = self.datum
            if d.poseHeatMaps is not None:
                self._pose_heatmaps = np.moveaxis(d.poseHeatMaps, 0, -1)
            else:
                size = d.netInputSizes[0]
                self._pose_heatmaps = np.zeros((size.y, size.x, d.poseKeypoints.shape[1]))
        return self._pose_heatmaps

    @property
    def pose_bounding_box(self) -> Optional[np.ndarray]:
        if self._pose_empty:
            return None
        if self._pose_bounding_box is None:
            self._pose_bounding_box = torso_boxes(self.pose_keypoints, TORSO_IDS)
        return self._pose_bounding_box

    @property
    def pose_bounding_box_int(self) -> Optional[np.ndarray]:
        if self._pose_empty:
            return None
        if self._pose_bounding_box_int is None and self.pose_bounding_box is not None:
            self._pose_bounding_box_int = integer_boxes(self.pose_bounding_box)
        return self._pose_bounding_box_int

    # ---- 3-D
    @staticmethod
    def get_3d_skeleton(skeleton: np.ndarray, depth_img: np.ndarray, intr_mat: Union[list, np.ndarray],
                        depth_scale: float = 1e-3, patch_offset: int = 2) -> np.ndarray:
        """One person: [J, 3] image joints -> [J, 3] camera-space joints (see :func:`lift_to_3d`)."""
        return lift_to_3d(skeleton, depth_img, intr_mat, depth_scale, patch_offset)

    def convert_to_3d(self, depth_image: np.ndarray, intr_mat: Union[list, np.ndarray],
                      depth_scale: float = 1e-3) -> None:
        if self.pose_empty:
            return
        self._pose_keypoints_3d = lift_to_3d(self.pose_keypoints, depth_image, intr_mat, depth_scale,
                                             self.patch_offset)

    # ---- writers
    @staticmethod
    def save_2d_skeleton(keypoints: Optional[np.ndarray], scores: Optional[np.ndarray], save_path: str) -> None:
        write_people_csv(save_path, keypoints, scores)

    def save_pose_keypoints(self, save_path: str) -> None:
        write_people_csv(save_path, self.pose_keypoints, self.pose_scores)

    def save_3d_pose_keypoints(self, save_path: str) -> None:
        write_people_csv(save_path, self.pose_keypoints_3d, self.pose_scores)

    # ---- drawing (NumPy rasteriser, posehip/draw.py; off the hot path)
    @staticmethod
    def _draw_text_on_skeleton_image(image: Optional[np.ndarray] = None,
                                     scores: Optional[np.ndarray] = None) -> np.ndarray:
        return image  # text rendering needs cv2.putText; omitted

    @staticmethod
    def _draw_depth_on_skeleton_image(image: Optional[np.ndarray] = None,
                                      depth: Optional[np.ndarray] = None) -> np.ndarray:
        if depth is None or image is None:
            return image
        g = np.clip(np.abs(depth.astype(np.float32)) * 0.065, 0, 255).astype(np.uint8)
        tint = np.stack([g // 3, g // 2, g], axis=-1).astype(np.float32)   # stand-in for COLORMAP_INFERNO
        return np.clip((image.astype(np.float32) + tint) * 0.7, 0, 255).astype(np.uint8)

    @staticmethod
    def _draw_bounding_box_on_skeleton_image(image: np.ndarray, boxes: Optional[np.ndarray]) -> np.ndarray:
        for bb in ([] if boxes is None else boxes):
            image = draw.rectangle(image, bb[0:2], bb[2:4], (0, 125, 0), 2)
        return image

    @staticmethod
    def _draw_tracking_bounding_box_image(image: np.ndarray, tracks: Optional[list] = None) -> np.ndarray:
        """Half-transparent green patch over the box of every activated track (``to_tlbr()`` or ``.tlbr``)."""
        tint = np.array((60, 130, 0), dtype=np.float32)
        for track in (tracks or []):
            if not track.is_activated:
                continue
            box = track.to_tlbr() if hasattr(track, "to_tlbr") else track.tlbr
            try:
                left, top, right, bottom = box
            except TypeError:
                continue
            win = (slice(int(np.floor(top)), int(np.ceil(bottom))), slice(int(np.floor(left)), int(np.ceil(right))))
            image[win] = np.clip((image[win] + tint) * 0.5 + 1.0, 0, 255).astype(np.uint8)
        return image

    def display(self, win_name: str, speed: int = 1000, scale: float = 1.0,
                depth_image: Optional[np.ndarray] = None, bounding_box: bool = False,
                tracks: Optional[list] = None, ori_image: Optional[np.ndarray] = None
                ) -> Tuple[bool, Optional[np.ndarray]]:
        """No window is opened (no cv2 here): returns (True, the composed image) for the caller to show."""
        def scaled(img):
            return draw.resize_nearest(img, int(img.shape[1] * scale), int(img.shape[0] * scale))
        canvas = self._draw_depth_on_skeleton_image(self.opencv_image, depth_image)
        if bounding_box:
            canvas = self._draw_bounding_box_on_skeleton_image(canvas, self.pose_bounding_box_int)
        if tracks is not None:
            canvas = self._draw_tracking_bounding_box_image(canvas, tracks)
        canvas = scaled(canvas)
        if ori_image is not None:
            canvas = np.concatenate([scaled(ori_image), canvas], axis=0)
        return True, canvas

    def save_skeleton_image(self, save_path: str, depth: Optional[np.ndarray] = None) -> None:
        from PIL import Image
        bgr = self._draw_depth_on_skeleton_image(self.opencv_image, depth)
        os.makedirs(os.path.dirname(save_path), exist_ok=True)
        Image.fromarray(np.ascontiguousarray(bgr[:, :, ::-1])).save(save_path)


_DEFAULT_PARAMS = {"model_folder": "/usr/local/src/openpose/models/", "model_pose": "BODY_25",
                   "net_resolution": "-1x368"}
_FORCED_PARAMS = {"disable_blending": True, "scale_number": 1, "body": 1}


class PyOpenPoseNative(PyOpenPoseNativeBase):
    """The estimator (interface of ``native/python/skeleton.py:373-514``) over :class:`Wrapper`."""

    def __init__(self, params: Optional[dict] = None, skel_thres: float = 0.0, max_true_body: int = 2,
                 patch_offset: int = 2) -> None:
        super().__init__(skel_thres, max_true_body, patch_offset)
        if params is None:
            params = dict(_DEFAULT_PARAMS)
        params.update(_FORCED_PARAMS)          # (in the caller's dict too, as the reference does)
        self.params, self.params_cout, self.params_cin = dict(params), dict(params), dict(params)
        self.opWrapper = Wrapper()
        self.configure(self.params, restart=False)
        self.filtered_skel = "0"
        self.reset()

    def reset(self) -> None:
        super().reset()
        self.filtered_skel = "0"

    def configure(self, params: dict = None, restart: bool = True) -> None:
        """Stop, re-configure and start the engine; ``None`` leaves it as it is."""
        if params is None:
            return
        if restart:
            self.opWrapper.stop()
        self.opWrapper.configure(params)
        self.opWrapper.start()

    def _process(self, image: np.ndarray, custom: Optional[List[np.ndarray]] = None, raw: bool = False) -> None:
        """One frame through the wrapper into ``self.datum``."""
        self.reset()
        self.datum.cvInputData = image
        if custom is not None:
            self.datum.customInputNetData = custom
        self.opWrapper.want_raw = raw
        try:
            self.opWrapper.emplaceAndPop([self.datum])
        finally:
            self.opWrapper.want_raw = False
            if custom is not None:
                self.datum.customInputNetData = []
        self._pose_empty = self.datum.poseScores is None

    def predict(self, image: np.ndarray) -> None:
        self._process(image)

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
        self._pose_empty = datum.poseScores is None

    def predict_hm(self, image: np.ndarray, hm_save_path: str) -> None:
        """Predict and dump the raw net output as a ``.float`` file: float32 [ndim, d0, d1, d2, d3, data...]
        (the header-prefixed write of the reference, :460-464; reader: :func:`read_heatmap_file`)."""
        self._process(image, raw=True)
        os.makedirs(os.path.dirname(hm_save_path), exist_ok=True)
        raw = self.datum.poseRawHeatMaps[0]
        header = np.asarray((raw.ndim,) + raw.shape, dtype=raw.dtype)
        np.concatenate([header, raw.reshape(-1)]).tofile(hm_save_path)

    def predict_from_hm(self, image: np.ndarray, heatmap: Optional[List[np.ndarray]] = None) -> None:
        """Decode-only: `heatmap` = [net output f32 [1, C, hs, ws]] is injected instead of running the network."""
        self._process(image, custom=heatmap if heatmap is not None else [])

    def filter_prediction(self) -> None:
        """Keep the ``max_true_body`` best people with a score of at least ``skel_thres``.  As in the reference
        (:483-514) the kept arrays go to ``_pose_keypoints`` / ``_pose_scores``, which no property reads: the public
        results stay unfiltered and only ``filtered_skel`` ("kept/total") reports the outcome."""
        if self.pose_empty:
            return
        scores = self.pose_scores
        keep = strongest_people(scores, self.max_true_body, self.skel_thres)
        if keep.size:
            self._pose_keypoints = self.pose_keypoints[keep]
            self._pose_scores = scores[keep]
        else:
            self.reset()
        self.filtered_skel = f"{keep.size}/{len(scores)}"

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


# argparse flag (openpose/native/python/args.py) -> OpenPose parameter
_FLAG_TO_PARAM = {"op_model_folder": "model_folder", "op_model_pose": "model_pose",
                  "op_net_resolution": "net_resolution", "op_heatmaps_add_parts": "heatmaps_add_parts",
                  "op_heatmaps_add_bkg": "heatmaps_add_bkg", "op_heatmaps_add_PAFs": "heatmaps_add_PAFs",
                  "op_heatmaps_scale": "heatmaps_scale"}
_POSEHIP_FLAGS = ("posehip_random_weights", "posehip_device", "posehip_batch")


class OpenPosePoseExtractor:
    """The argparse-driven facade (interface of ``native/python/skeleton.py:517-608``)."""

    def __init__(self, args: argparse.Namespace) -> None:
        params = {param: getattr(args, flag) for flag, param in _FLAG_TO_PARAM.items()}
        params["render_pose"] = -1 if args.op_display > 0 else 0
        params.update({k: getattr(args, k) for k in _POSEHIP_FLAGS if getattr(args, k, None) is not None})
        self.pyop = PyOpenPoseNative(params, args.op_skel_thres, args.op_max_true_body, args.op_patch_offset)

    def _after(self, kpt_save_path: Optional[str]) -> None:
        self.pyop.filter_prediction()
        if kpt_save_path is not None:
            self.pyop.save_pose_keypoints(kpt_save_path)

    def predict(self, image: np.ndarray, depth: np.ndarray, kpt_save_path: Optional[str] = None,
                skel_image_save_path: Optional[str] = None) -> None:
        self.pyop.predict(image)
        self._after(kpt_save_path)
        if skel_image_save_path is not None:
            self.pyop.save_skeleton_image(skel_image_save_path, depth)

    def predict_hm(self, image: np.ndarray, hm_save_path: str) -> None:
        self.pyop.predict_hm(image, hm_save_path)

    def predict_from_hm(self, image: np.ndarray, heatmap: Optional[List[np.ndarray]] = None,
                        kpt_save_path: Optional[str] = None) -> None:
        self.pyop.predict_from_hm(image, heatmap)
        self._after(kpt_save_path)

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
        """(True, composed image) -- or (True, None) for a scale of 0 or less; nothing is shown on screen."""
        if not scale > 0:
            return True, None
        if self.pyop.datum.poseScores is not None:
            return self.pyop.display(win_name, scale=scale, speed=speed, bounding_box=bounding_box, tracks=tracks,
                                     ori_image=image)
        canvas = self.pyop.opencv_image
        if image is not None:
            top = draw.resize_nearest(image, int(image.shape[1] * scale), int(image.shape[0] * scale))
            canvas = np.concatenate([top, canvas], axis=0)
        return True, canvas
