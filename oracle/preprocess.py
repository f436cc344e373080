"""ORACLE (test infrastructure only -- never imported by the product path).

CPU restatement of the K1 pre-processing the reference delegates to OpenCV:
``cv2.dnn.blobFromImage(image, 1/255, (Wn, Hn), (0,0,0), swapRB=False, crop=False)``
(openpose/opencv/skeleton.py:66-73) = ``cv2.resize`` INTER_LINEAR on uint8 followed by
x * (1/255) in fp32, HWC -> NCHW with the BGR order kept.

OpenCV itself is not installed anywhere this runs: the uint8 resize below
restates the published OpenCV 4.5 algorithm (``resizeGeneric_`` with the 11-bit
fixed-point ``HResizeLinear`` / ``VResizeLinear`` pair: INTER_RESIZE_COEF_BITS = 11,
horizontal pass into int32, vertical pass ``((b0*(S0>>4))>>16) + ((b1*(S1>>4))>>16) + 2) >> 2``).
PARITY UNPINNED against a real cv2 build; pinned only by its own known answers
(tests/test_oracle_v1.py) and by the committed real-image fixture it generated.
"""
import numpy as np


def _coords(n_dst, n_src, clamp_reset):
    d = np.arange(n_dst, dtype=np.float64)
    f = ((d + 0.5) * (n_src / n_dst) - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int32)
    f = (f - s).astype(np.float32)
    if clamp_reset:  # the x pass zeroes the fraction at both borders, the y pass clamps the rows
        lo, hi = s < 0, s >= n_src - 1
        f[lo] = 0
        s[lo] = 0
        f[hi] = 0
        s[hi] = n_src - 1
    c0 = np.rint((np.float32(1) - f) * np.float32(2048)).astype(np.int32)
    c1 = np.rint(f * np.float32(2048)).astype(np.int32)
    return s, c0, c1


def cv_resize_u8(img, W, H):
    """uint8 [B,H0,W0,C] -> uint8 [B,H,W,C], cv2.resize INTER_LINEAR restated."""
    B, H0, W0, _ = img.shape
    if (H0, W0) == (H, W):
        return img.copy()
    sx, a0, a1 = _coords(W, W0, True)
    sy, b0, b1 = _coords(H, H0, False)
    sx1 = np.minimum(sx + 1, W0 - 1)
    sy0, sy1 = np.clip(sy, 0, H0 - 1), np.clip(sy + 1, 0, H0 - 1)
    im = img.astype(np.int32)
    rows = im[:, :, sx] * a0[None, None, :, None] + im[:, :, sx1] * a1[None, None, :, None]
    S0, S1 = rows[:, sy0], rows[:, sy1]
    d = (((b0[None, :, None, None] * (S0 >> 4)) >> 16) + ((b1[None, :, None, None] * (S1 >> 4)) >> 16) + 2) >> 2
    return np.clip(d, 0, 255).astype(np.uint8)


def blob_from_image(img, W, H):
    """uint8 [B,H0,W0,3] BGR -> float32 [B,3,H,W] (scalefactor 1/255, no mean, no swap, no crop)."""
    r = cv_resize_u8(img, W, H).astype(np.float32) * np.float32(1.0 / 255.0)
    return np.ascontiguousarray(r.transpose(0, 3, 1, 2))
