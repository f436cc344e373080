"""Minimal NumPy rasteriser (cv2 is not available): thick lines, discs, boxes.

Only the drawing the reference's estimator classes do with cv2 is covered
(``openpose/opencv/skeleton.py:156-227``, ``openpose/native/python/skeleton.py:215-291``);
anti-aliasing and text are omitted.  Not on the hot path.
"""
import numpy as np


def disc(img, center, radius, color):
    h, w = img.shape[:2]
    cx, cy = int(round(float(center[0]))), int(round(float(center[1])))
    x0, x1 = max(0, cx - radius), min(w, cx + radius + 1)
    y0, y1 = max(0, cy - radius), min(h, cy + radius + 1)
    if x0 >= x1 or y0 >= y1:
        return img
    yy, xx = np.mgrid[y0:y1, x0:x1]
    m = (xx - cx) ** 2 + (yy - cy) ** 2 <= radius * radius
    img[y0:y1, x0:x1][m] = color
    return img


def line(img, p0, p1, color, thickness=1):
    x0, y0 = float(p0[0]), float(p0[1])
    x1, y1 = float(p1[0]), float(p1[1])
    n = int(max(abs(x1 - x0), abs(y1 - y0))) + 1
    r = max(0, thickness // 2)
    for t in np.linspace(0.0, 1.0, n):
        disc(img, (x0 + t * (x1 - x0), y0 + t * (y1 - y0)), r, color)
    return img


def rectangle(img, tl, br, color, thickness=1):
    x0, y0, x1, y1 = int(tl[0]), int(tl[1]), int(br[0]), int(br[1])
    line(img, (x0, y0), (x1, y0), color, thickness)
    line(img, (x1, y0), (x1, y1), color, thickness)
    line(img, (x1, y1), (x0, y1), color, thickness)
    line(img, (x0, y1), (x0, y0), color, thickness)
    return img


def resize_nearest(img, width, height):
    h, w = img.shape[:2]
    ys = np.minimum((np.arange(height) * h / max(height, 1)).astype(int), h - 1)
    xs = np.minimum((np.arange(width) * w / max(width, 1)).astype(int), w - 1)
    return img[ys][:, xs]


_PALETTE = [(255, 0, 85), (255, 0, 0), (255, 85, 0), (255, 170, 0), (255, 255, 0), (170, 255, 0), (85, 255, 0),
            (0, 255, 0), (255, 0, 0), (0, 255, 85), (0, 255, 170), (0, 255, 255), (0, 170, 255), (0, 85, 255),
            (0, 0, 255), (255, 0, 170), (170, 0, 255), (255, 0, 255), (85, 0, 255), (0, 0, 255), (0, 0, 255),
            (0, 0, 255), (0, 255, 255), (0, 255, 255), (0, 255, 255)]


def render_pose(image, keypoints, pairs, threshold=0.05):
    """Skeleton sticks + joints on `image` (u8 HWC BGR), in place."""
    if keypoints is None:
        return image
    for person in keypoints:
        for li, (a, b) in enumerate(pairs):
            if person[a, 2] > threshold and person[b, 2] > threshold:
                c = _PALETTE[b % len(_PALETTE)][::-1]
                line(image, person[a, :2], person[b, :2], c, 3)
        for k in range(person.shape[0]):
            if person[k, 2] > threshold:
                disc(image, person[k, :2], 3, _PALETTE[k % len(_PALETTE)][::-1])
    return image
