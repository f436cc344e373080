"""File readers / save-path builder of the reference's pipeline callers
(``openpose/native/python/utils_rs.py``): same names, arguments,
return layouts.  ``cv2`` is not used: images are read with PIL and the YUYV
conversion is restated in NumPy (integer BT.601, OpenCV's constants).
"""
import csv
import json
import os
from typing import Optional

import numpy as np


def get_rs_sensor_dir(base_path: str, sensor: str) -> dict:
    """{device: {timestamp: base/device/timestamp/sensor}} (``utils_rs.py:13-20``)."""
    out = {}
    for device in sorted(os.listdir(base_path)):
        out[device] = {ts: os.path.join(base_path, device, ts, sensor)
                       for ts in sorted(os.listdir(os.path.join(base_path, device)))}
    return out


def _get_brg_from_yuv(data_array: np.ndarray) -> np.ndarray:
    """u16 YUYV words (low byte Y, high byte U or V alternating) -> BGR8 [N,3]
    (``utils_rs.py:25-36``: ``cv2.cvtColor(.., COLOR_YUV2BGR_YUYV)``).

    Restated from OpenCV's published integer BT.601 conversion (20-bit fixed point); pixels are
    paired along the flat sequence (Y0 U | Y1 V).  Parity unpinned vs real cv2 (absent here)."""
    d = np.asarray(data_array).reshape(-1)
    if d.size % 2:
        raise ValueError("YUYV data needs an even number of pixels")
    y = (d & 0xFF).astype(np.int64)
    c = (d >> 8).astype(np.int64)
    u = np.repeat(c[0::2], 2) - 128
    v = np.repeat(c[1::2], 2) - 128
    yy = np.maximum(y - 16, 0) * 1220542
    half = 1 << 19
    r = (yy + half + 1673527 * v) >> 20
    g = (yy + half - 852492 * v - 409993 * u) >> 20
    b = (yy + half + 2116026 * u) >> 20
    return np.clip(np.stack([b, g, r], axis=1), 0, 255).astype(np.uint8)


def read_color_file(filename: str, fileformat: Optional[str] = None) -> np.ndarray:
    """Colour frame as BGR u8 (``utils_rs.py:41-62``): image files [H,W,3]; ``.bin`` raw bytes
    (flat, the caller reshapes) in bgr8/rgb8/yuyv; anything else via ``np.load`` (u8 as is,
    u16 = YUYV)."""
    if filename.endswith(('.png', '.jpeg', '.jpg')):
        from PIL import Image
        return np.array(Image.open(filename))[:, :, ::-1].copy()
    if filename.endswith('.bin'):
        fmt = fileformat.lower()
        if fmt in ('bgr8', 'rgb8'):
            return np.fromfile(filename, np.uint8)
        if fmt == 'yuyv':
            return _get_brg_from_yuv(np.fromfile(filename, np.uint16))
        raise ValueError("Unknown data type :", fileformat)
    image = np.load(filename)
    if image.dtype == np.dtype(np.uint8):
        return image
    if image.dtype == np.dtype(np.uint16):
        return _get_brg_from_yuv(image.reshape(-1))
    raise ValueError("Unknown data type :", image.dtype)


def read_depth_file(filename: str) -> np.ndarray:
    """Raw u16 depth, flat (``utils_rs.py:67-73``)."""
    return np.fromfile(filename, np.uint16)


def _stream_row(row) -> dict:
    return {'width': int(row[0]), 'height': int(row[1]), 'ppx': float(row[2]), 'ppy': float(row[3]),
            'fx': float(row[4]), 'fy': float(row[5]), 'model': row[6],
            'coeffs': [float(i) for i in row[7:12]], 'format': row[12], 'fps': int(row[13])}


def read_calib_file(calib_file: str) -> dict:
    """RealSense calibration (``utils_rs.py:78-123``): JSON as is; CSV rows = colour stream,
    depth stream, colour<-depth extrinsics (9 rotation + 3 translation), depth scale/baseline."""
    calib = {}
    if calib_file.endswith('.json'):
        with open(calib_file, 'r') as f:
            calib = json.load(f)
    elif calib_file.endswith('.csv'):
        with open(calib_file, 'r') as f:
            for n, row in enumerate(csv.reader(f, delimiter=',')):
                if n == 0:
                    calib['color'] = _stream_row(row)
                elif n == 1:
                    calib['depth'] = _stream_row(row)
                elif n == 2:
                    calib['T_color_depth'] = {'rotation': [float(i) for i in row[0:9]],
                                              'translation': [float(i) for i in row[9:12]]}
                elif n == 3:
                    calib['depth']['depth_scale'] = float(row[0])
                    calib['depth']['depth_baseline'] = float(row[1])
    return calib


def prepare_save_paths(skeleton_folder: Optional[str] = None, prefix: str = '', save_skel: bool = False,
                       save_skel_image: bool = False, save_3d_skel: bool = False) -> dict:
    """Output paths of one frame (``utils_rs.py:126-154``); creates the folder."""
    if skeleton_folder is None:
        assert not save_skel and not save_skel_image and not save_3d_skel
    else:
        os.makedirs(skeleton_folder, exist_ok=True)

    def path(flag, suffix):
        return os.path.join(skeleton_folder, f'{prefix}{suffix}') if flag else None

    return {'skel_save_path': path(save_skel, '_skeleton.csv'),
            '3dskel_save_path': path(save_3d_skel, '_skeleton3d.csv'),
            'skel_image_save_path': path(save_skel_image, '_skeleton.jpg')}
