"""Host helpers the reference's pipeline callers import from
``openpose/native/python/utils.py``: same names, arguments and
results; nothing here touches the GPU.
"""
import argparse
import time

import numpy as np


def dict_check(x: dict) -> bool:
    """True when every value's ``.state`` is set (``utils.py:7-12``)."""
    return sum(1 for v in x.values() if v.state) == len(x)


class Timer:
    """``with Timer(text, enable, printout) as t: ...; t.duration`` (``utils.py:15-41``).
    Wall clock (``time.time``); ``duration`` stays None when disabled."""

    def __init__(self, text: str = '', enable: bool = True, printout: bool = False) -> None:
        self._text = text
        self._enable = enable
        self._printout = printout
        self._t_start = None
        self._duration = None

    @property
    def duration(self):
        return self._duration

    def __enter__(self):
        if self._enable:
            self._t_start = time.time()
        return self

    def __exit__(self, *exc) -> None:
        if not self._enable:
            return
        self._duration = time.time() - self._t_start
        if self._printout:
            print(f"[INFO] : Timer >>> {self._text: <40} >>> {self._duration:.6f} s")


class Error:
    """State/counter pair used by the RealSense loops (``utils.py:44-51``)."""

    def __init__(self):
        self.reset()

    def reset(self):
        self.state = False
        self.counter = 0


_TRUE = ('yes', 'true', 't', 'y', '1')
_FALSE = ('no', 'false', 'f', 'n', '0')


def str2bool(v: str) -> bool:
    """argparse ``type=`` for boolean flags (``utils.py:54-71``)."""
    s = v.lower()
    if s in _TRUE:
        return True
    if s in _FALSE:
        return False
    raise argparse.ArgumentTypeError('Boolean value expected.')


def cosine_distance(a: np.ndarray, b: np.ndarray, data_is_normalized: bool = False) -> np.ndarray:
    """Pairwise ``1 - cos`` between the rows of a [N,M] and b [L,M] -> [N,L] (``utils.py:75-97``)."""
    if not data_is_normalized:
        a = np.asarray(a) / np.linalg.norm(a, axis=1, keepdims=True)
        b = np.asarray(b) / np.linalg.norm(b, axis=1, keepdims=True)
    return 1. - np.dot(a, b.T)


def resize_tensor_indices(h: int, w: int, x: int, y: int):
    """Source (row, column) indices that ``resize_tensor`` reads for output rows 1..y-1 and
    columns 1..x-1 (row 0 and column 0 of its output stay zero).  Same float arithmetic as the
    reference's loop: ``1 + int(i / (y / max(h-1, 1)))``."""
    x_scale = x / max(w - 1, 1)
    y_scale = y / max(h - 1, 1)
    rows = 1 + (np.arange(y - 1) / y_scale).astype(np.int64)
    cols = 1 + (np.arange(x - 1) / x_scale).astype(np.int64)
    return rows, cols


def resize_tensor(image: np.ndarray, x: int, y: int) -> np.ndarray:
    """The reference's nearest-neighbour-like resampler (``utils.py:101-110``), vectorised:
    out [y, x, c] float64, ``out[i+1, j+1] = image[1 + int(i/yScale), 1 + int(j/xScale)]``,
    first row and column zero.  Raises IndexError like the reference when the input has a
    single row or column (index 1 does not exist)."""
    h, w, c = image.shape
    out = np.zeros([y, x, c])
    if y > 1 and x > 1:
        rows, cols = resize_tensor_indices(h, w, x, y)
        if rows.max() >= h or cols.max() >= w:
            raise IndexError(f"index out of bounds for a {h}x{w} tensor")
        out[1:, 1:] = image[rows[:, None], cols[None, :]]
    return out


def get_color(idx):
    idx = idx * 3
    return ((37 * idx) % 255, (17 * idx) % 255, (29 * idx) % 255)


_COLOR_POOL = [(0, 0, 255), (0, 255, 0), (255, 0, 0), (255, 255, 0),
               (0, 255, 255), (0, 165, 255), (255, 0, 255), (255, 255, 255)]  # BGR


def get_color_fast(idx):
    return _COLOR_POOL[idx % 8]
