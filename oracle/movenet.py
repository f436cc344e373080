"""ORACLE (test infrastructure only -- never imported by the product path).

SURVEY 8f-4 / BASELINE configs[3]: the MoveNet steps that exist in the reference tree, restated, plus a
MultiPose head decode that does NOT exist there.

In tree (movenet/utils/image_sequence_crop.py, plain Python around a remote TF-Hub model):
* ``init_crop_region``            :15-40
* ``torso_visible``               :43-56
* ``determine_torso_and_body_range`` :59-91
* ``determine_crop_region``       :94-150
* ``crop_and_resize``             :153-159 -> ``tf.image.crop_and_resize`` (bilinear, extrapolation 0), then
                                  ``tf.cast(.., tf.int32)`` (:171); restated from TensorFlow's published kernel
                                  (crop_and_resize_op.cc); **unpinned vs real TensorFlow** (absent offline)
* ``update_keypoints``            the coordinate update loop of ``run_inference`` :178-186
The module imports tensorflow at its top and cannot be imported here, so these are pinned by hand-computed
cases only (tests/test_oracle_movenet.py).  Arithmetic on ``keypoints[...] * image_height`` is float64
(NumPy < 2 promotion of a float32 scalar and a Python int, the reference's TF 2.5 era), stored back to float32.

NOT in tree: ``multipose_decode``.  The reference only ever calls the SinglePose models, whose decode is inside
the TF-Hub graph.  BASELINE configs[3] asks for "MultiPose Lightning 256x256 heatmap/offset decoding, batch 64";
this is the decode as the MoveNet model card / blog describe it (centre heat map -> top-K local maxima ->
keypoint regression from the centre -> refinement on the keypoint heat maps weighted by inverse distance ->
local offsets; box from the size head), written down HERE as the specification.  **Parity unpinned**: there is
no reference output for it anywhere.
"""
import numpy as np

KEYPOINT_DICT = {
    'nose': 0, 'left_eye': 1, 'right_eye': 2, 'left_ear': 3, 'right_ear': 4, 'left_shoulder': 5,
    'right_shoulder': 6, 'left_elbow': 7, 'right_elbow': 8, 'left_wrist': 9, 'right_wrist': 10, 'left_hip': 11,
    'right_hip': 12, 'left_knee': 13, 'right_knee': 14, 'left_ankle': 15, 'right_ankle': 16}
MIN_CROP_KEYPOINT_SCORE = 0.2


def init_crop_region(image_height, image_width):
    if image_width > image_height:
        box_height = image_width / image_height
        box_width = 1.0
        y_min = (image_height / 2 - image_width / 2) / image_height
        x_min = 0.0
    else:
        box_height = 1.0
        box_width = image_height / image_width
        y_min = 0.0
        x_min = (image_width / 2 - image_height / 2) / image_width
    return {'y_min': y_min, 'x_min': x_min, 'y_max': y_min + box_height, 'x_max': x_min + box_width,
            'height': box_height, 'width': box_width}


def torso_visible(keypoints):
    s = lambda n: float(keypoints[0, 0, KEYPOINT_DICT[n], 2])  # noqa: E731
    return ((s('left_hip') > MIN_CROP_KEYPOINT_SCORE or s('right_hip') > MIN_CROP_KEYPOINT_SCORE) and
            (s('left_shoulder') > MIN_CROP_KEYPOINT_SCORE or s('right_shoulder') > MIN_CROP_KEYPOINT_SCORE))


def determine_torso_and_body_range(keypoints, target_keypoints, center_y, center_x):
    max_torso_yrange = max_torso_xrange = 0.0
    for joint in ('left_shoulder', 'right_shoulder', 'left_hip', 'right_hip'):
        max_torso_yrange = max(max_torso_yrange, abs(center_y - target_keypoints[joint][0]))
        max_torso_xrange = max(max_torso_xrange, abs(center_x - target_keypoints[joint][1]))
    max_body_yrange = max_body_xrange = 0.0
    for joint, idx in KEYPOINT_DICT.items():
        if float(keypoints[0, 0, idx, 2]) < MIN_CROP_KEYPOINT_SCORE:
            continue
        max_body_yrange = max(max_body_yrange, abs(center_y - target_keypoints[joint][0]))
        max_body_xrange = max(max_body_xrange, abs(center_x - target_keypoints[joint][1]))
    return [max_torso_yrange, max_torso_xrange, max_body_yrange, max_body_xrange]


def determine_crop_region(keypoints, image_height, image_width):
    target = {j: [float(keypoints[0, 0, i, 0]) * image_height, float(keypoints[0, 0, i, 1]) * image_width]
              for j, i in KEYPOINT_DICT.items()}
    if not torso_visible(keypoints):
        return init_crop_region(image_height, image_width)
    center_y = (target['left_hip'][0] + target['right_hip'][0]) / 2
    center_x = (target['left_hip'][1] + target['right_hip'][1]) / 2
    ty, tx, by, bx = determine_torso_and_body_range(keypoints, target, center_y, center_x)
    half = max(tx * 1.9, ty * 1.9, by * 1.2, bx * 1.2)
    half = min(half, max(center_x, image_width - center_x, center_y, image_height - center_y))
    corner = [center_y - half, center_x - half]
    if half > max(image_width, image_height) / 2:
        return init_crop_region(image_height, image_width)
    length = half * 2
    return {'y_min': corner[0] / image_height, 'x_min': corner[1] / image_width,
            'y_max': (corner[0] + length) / image_height, 'x_max': (corner[1] + length) / image_width,
            'height': (corner[0] + length) / image_height - corner[0] / image_height,
            'width': (corner[1] + length) / image_width - corner[1] / image_width}


def crop_and_resize(image, crop_region, crop_size):
    """image u8/float [H,W,C] -> int32 [ch, cw, C]: tf.image.crop_and_resize (bilinear, extrapolation_value 0)
    of the one box [y_min, x_min, y_max, x_max], then tf.cast(int32) (truncation).  float32 as in the TF kernel."""
    f = np.float32
    img = np.asarray(image).astype(f)
    H, W, C = img.shape
    ch, cw = crop_size
    y1, x1, y2, x2 = (f(crop_region[k]) for k in ('y_min', 'x_min', 'y_max', 'x_max'))
    hs = (y2 - y1) * f(H - 1) / f(ch - 1) if ch > 1 else f(0)
    ws = (x2 - x1) * f(W - 1) / f(cw - 1) if cw > 1 else f(0)
    out = np.zeros((ch, cw, C), f)
    for y in range(ch):
        in_y = y1 * f(H - 1) + f(y) * hs if ch > 1 else f(0.5) * (y1 + y2) * f(H - 1)
        if in_y < 0 or in_y > H - 1:
            continue
        top, bot = int(np.floor(in_y)), int(np.ceil(in_y))
        y_lerp = in_y - f(top)
        for x in range(cw):
            in_x = x1 * f(W - 1) + f(x) * ws if cw > 1 else f(0.5) * (x1 + x2) * f(W - 1)
            if in_x < 0 or in_x > W - 1:
                continue
            left, right = int(np.floor(in_x)), int(np.ceil(in_x))
            x_lerp = in_x - f(left)
            t = img[top, left] + (img[top, right] - img[top, left]) * x_lerp
            b = img[bot, left] + (img[bot, right] - img[bot, left]) * x_lerp
            out[y, x] = t + (b - t) * y_lerp
    return out.astype(np.int32)


def update_keypoints(keypoints_with_scores, crop_region, image_height, image_width):
    """[1,1,17,3] f32 (y, x, score) in crop coordinates -> image coordinates, in place semantics of :178-186."""
    k = np.array(keypoints_with_scores, np.float32)
    for idx in range(17):
        k[0, 0, idx, 0] = (crop_region['y_min'] * image_height +
                           crop_region['height'] * image_height * float(k[0, 0, idx, 0])) / image_height
        k[0, 0, idx, 1] = (crop_region['x_min'] * image_width +
                           crop_region['width'] * image_width * float(k[0, 0, idx, 1])) / image_width
    return k


# ---------------------------------------------------------------------------------------------------------
def multipose_decode(center, kpt_heat, kpt_regr, kpt_offs, box_size, max_people=6, min_center=0.0):
    """Specification (not a restatement of in-tree code, see the module header).

    Heads of one frame on a G x G grid (G = 64 for a 256 x 256 input), float32, scores already in [0,1]:
      center   [G,G]        person-centre heat map
      kpt_heat [G,G,17]     keypoint heat maps
      kpt_regr [G,G,34]     (dy, dx) per keypoint from a centre cell to the keypoint, in grid cells
      kpt_offs [G,G,34]     (dy, dx) sub-cell offset at a keypoint's cell, in grid cells
      box_size [G,G,2]      (height, width) of the person box at a centre cell, in grid cells
    -> [max_people, 56] f32: 17 x (y, x, score) normalised to [0,1], then (ymin, xmin, ymax, xmax, score);
       rows sorted by centre score (ties: lower raster index first); unused rows zero.

    1. centres: cells that are >= all 8 neighbours (out-of-range neighbours ignored) and > min_center;
       the max_people highest.
    2. coarse keypoint j of a person = centre cell + kpt_regr[centre, 2j:2j+2].
    3. refine: cell (y, x) maximising kpt_heat[y, x, j] / (distance((y, x), coarse) + 1.8) over the whole grid
       (first in raster order on ties); keypoint = cell + kpt_offs[cell, 2j:2j+2]; score = kpt_heat[cell, j].
    4. box = centre +- box_size[centre] / 2; everything divided by G; box score = centre score.
    """
    f = np.float32
    G = center.shape[0]
    c = np.asarray(center, f)
    pad = np.full((G + 2, G + 2), -np.inf, f)
    pad[1:-1, 1:-1] = c
    is_max = np.ones((G, G), bool)
    for dy in (-1, 0, 1):
        for dx in (-1, 0, 1):
            if dy or dx:
                is_max &= c >= pad[1 + dy:G + 1 + dy, 1 + dx:G + 1 + dx]
    cand = np.flatnonzero((is_max & (c > f(min_center))).reshape(-1))
    order = sorted(cand.tolist(), key=lambda i: (-float(c.reshape(-1)[i]), i))[:max_people]
    out = np.zeros((max_people, 56), f)
    yy, xx = np.mgrid[0:G, 0:G]
    yy, xx = yy.astype(f), xx.astype(f)
    for p, idx in enumerate(order):
        cy, cx = divmod(idx, G)
        for j in range(17):
            ry = f(cy) + kpt_regr[cy, cx, 2 * j]
            rx = f(cx) + kpt_regr[cy, cx, 2 * j + 1]
            dy_, dx_ = yy - ry, xx - rx
            dist = np.sqrt(dy_ * dy_ + dx_ * dx_).astype(f)
            w = (kpt_heat[:, :, j].astype(f) / (dist + f(1.8))).astype(f)
            k = int(np.argmax(w.reshape(-1)))
            ky, kx = divmod(k, G)
            out[p, 3 * j] = (f(ky) + kpt_offs[ky, kx, 2 * j]) / f(G)
            out[p, 3 * j + 1] = (f(kx) + kpt_offs[ky, kx, 2 * j + 1]) / f(G)
            out[p, 3 * j + 2] = kpt_heat[ky, kx, j]
        h2, w2 = box_size[cy, cx, 0] * f(0.5), box_size[cy, cx, 1] * f(0.5)
        out[p, 51] = (f(cy) - h2) / f(G)
        out[p, 52] = (f(cx) - w2) / f(G)
        out[p, 53] = (f(cy) + h2) / f(G)
        out[p, 54] = (f(cx) + w2) / f(G)
        out[p, 55] = c[cy, cx]
    return out
