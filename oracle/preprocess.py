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


# ---------------------------------------------------------------------------------------------------------------
# Native flavour: OpenPose's CvMatToOpInput inside emplaceAndPop (openpose/native/python/skeleton.py:444-447; the
# inverse, (x + 0.5) * 255, is at tutorial_api_python/08_heatmaps_from_image.py:70).  Upstream OpenPose 1.7.0,
# restated from the published source (the fork the reference builds is not under /root/reference):
#   ScaleAndSizeExtractor::extract   net size from "-1x368": 16 * positiveIntRound(1/16.f * 368 * W0 / (float)H0)
#   resizeGetScaleFactor             s = min((Wn-1)/(W0-1), (Hn-1)/(H0-1))   (double)
#   resizeFixedAspectRatio           cv::warpAffine(src, dst, [[s,0,0],[0,s,0]], (Wn,Hn),
#                                    s > 1 ? INTER_CUBIC : INTER_AREA, BORDER_CONSTANT, 0)
#   uCharCvMatToFloatPtr(normalize=1)   x / 256.f - 0.5f, HWC -> CHW
# and OpenCV 4.5 imgwarp.cpp for cv::warpAffine on 8UC3: INTER_AREA is replaced by INTER_LINEAR; the forward matrix
# is inverted in double; source coordinates are fixed point, AB_BITS = 10, quantised to 1/32 px (INTER_BITS = 5)
# with round_delta = 16; remap's interpolation weights are 15-bit integers from the BilinearTab_i / BicubicTab_i
# tables (initInterTab2D, cubic A = -0.75, the "make the weights sum to 32768" fix-up included); the result is
# (sum + 16384) >> 15 saturated to u8; taps outside the image contribute the border value 0.
# PARITY UNPINNED against a real cv2 / pyopenpose (neither is installable here).
# ---------------------------------------------------------------------------------------------------------------
f32 = np.float32


def op_net_input_size(net_resolution, image_w, image_h):
    """ScaleAndSizeExtractor::extract: "WxH" with one side <= 0 = keep the aspect ratio, multiple of 16."""
    w, h = (int(v) for v in net_resolution.lower().split("x"))
    if w <= 0 and h <= 0:
        raise ValueError("net_resolution: width and height cannot both be -1")
    if w <= 0:
        w = 16 * int(f32(f32(f32(f32(1) / f32(16)) * f32(h)) * f32(image_w)) / f32(image_h) + f32(0.5))
    elif h <= 0:
        h = 16 * int(f32(f32(f32(f32(1) / f32(16)) * f32(w)) * f32(image_h)) / f32(image_w) + f32(0.5))
    return w, h


def op_scale_factor(W0, H0, Wn, Hn):
    """resizeGetScaleFactor (double)."""
    return min((Wn - 1) / float(W0 - 1), (Hn - 1) / float(H0 - 1))


def _cubic_coeffs(x):
    """interpolateCubic, fp32, A = -0.75."""
    A = f32(-0.75)
    x = f32(x)
    one = f32(1)
    c0 = f32(f32(f32(f32(f32(A * f32(x + one)) - f32(f32(5) * A)) * f32(x + one)) + f32(f32(8) * A)) * f32(x + one)) - f32(f32(4) * A)
    c1 = f32(f32(f32(f32(f32(A + f32(2)) * x) - f32(A + f32(3))) * x) * x) + one
    ix = f32(one - x)
    c2 = f32(f32(f32(f32(f32(A + f32(2)) * ix) - f32(A + f32(3))) * ix) * ix) + one
    c3 = f32(f32(f32(one - c0) - c1) - c2)
    return [f32(c0), f32(c1), f32(c2), f32(c3)]


def _inter_tab_2d(cubic):
    """initInterTab2D(..., fixpt=true): int weights [32*32][k][k], k = 4 (cubic) or 2 (linear)."""
    k = 4 if cubic else 2
    scale = f32(1.0 / 32)
    tab1 = []
    for i in range(32):
        x = f32(f32(i) * scale)
        tab1.append(_cubic_coeffs(x) if cubic else [f32(f32(1) - x), x])
    out = np.zeros((32 * 32 + 1, k, k), np.int32)      # one spare entry: the fix-up of k = 2 peeks past the current one
    flat = out.reshape(-1)
    for i in range(32):
        for j in range(32):
            base = (i * 32 + j) * k * k
            isum = 0
            for k1 in range(k):
                vy = tab1[i][k1]
                for k2 in range(k):
                    v = f32(vy * tab1[j][k2])
                    w = int(np.clip(np.rint(f32(v * f32(32768))), -32768, 32767))   # saturate_cast<short>(float)
                    flat[base + k1 * k + k2] = w
                    isum += w
            if isum != 32768:
                diff = isum - 32768
                h = k // 2
                Mk1 = Mk2 = mk1 = mk2 = h
                for k1 in range(h, h + 2):
                    for k2 in range(h, h + 2):
                        # (for k = 2 this window leaves the entry, exactly as the library's loop does; the entries
                        # after the current one are still zero when it runs)
                        if flat[base + k1 * k + k2] < flat[base + mk1 * k + mk2]:
                            mk1, mk2 = k1, k2
                        elif flat[base + k1 * k + k2] > flat[base + Mk1 * k + Mk2]:
                            Mk1, Mk2 = k1, k2
                if diff < 0:
                    flat[base + Mk1 * k + Mk2] -= diff
                else:
                    flat[base + mk1 * k + mk2] -= diff
    return out[:32 * 32]


_TABS = {}


def warp_affine_scale_u8(img, s, Wn, Hn):
    """cv::warpAffine(img, [[s,0,0],[0,s,0]], (Wn,Hn), s > 1 ? INTER_CUBIC : INTER_LINEAR, BORDER_CONSTANT, 0) on
    uint8 [B,H0,W0,C] -> uint8 [B,Hn,Wn,C]."""
    B, H0, W0, C = img.shape
    cubic = s > 1.0
    if cubic not in _TABS:
        _TABS[cubic] = _inter_tab_2d(cubic)
    tab = _TABS[cubic]
    k = 4 if cubic else 2
    # inverse of the forward matrix, as cv::warpAffine computes it
    D = s * s
    D = 1.0 / D if D != 0 else 0.0
    a11 = s * D
    xs = np.arange(Wn, dtype=np.float64)
    ys = np.arange(Hn, dtype=np.float64)
    adelta = np.rint(a11 * xs * 1024.0).astype(np.int64)             # saturate_cast<int>(M[0]*x*AB_SCALE)
    X = (16 + adelta) >> 5                                            # X0 = 0 + round_delta
    Y = (np.rint((a11 * ys + 0.0) * 1024.0).astype(np.int64) + 16) >> 5
    sx, ax = (X >> 5) - (1 if cubic else 0), X & 31
    sy, ay = (Y >> 5) - (1 if cubic else 0), Y & 31
    w = tab[(ay[:, None] * 32 + ax[None, :])]                          # [Hn, Wn, k, k]
    acc = np.zeros((B, Hn, Wn, C), np.int64)
    src = img.astype(np.int64)
    for i in range(k):
        yy = sy + i
        oky = (yy >= 0) & (yy < H0)
        yc = np.clip(yy, 0, H0 - 1)
        for j in range(k):
            xx = sx + j
            okx = (xx >= 0) & (xx < W0)
            xc = np.clip(xx, 0, W0 - 1)
            tap = src[:, yc][:, :, xc]                                # [B, Hn, Wn, C]
            m = (oky[:, None] & okx[None, :])[None, :, :, None]
            acc += np.where(m, tap, 0) * w[None, :, :, i, j, None]
    return np.clip((acc + 16384) >> 15, 0, 255).astype(np.uint8)


def op_resize_fixed_aspect(img, Wn, Hn):
    """resizeFixedAspectRatio: uint8 [B,H0,W0,3] -> uint8 [B,Hn,Wn,3]."""
    B, H0, W0, _ = img.shape
    s = op_scale_factor(W0, H0, Wn, Hn)
    if s == 1.0 and (W0, H0) == (Wn, Hn):
        return img.copy()
    return warp_affine_scale_u8(img, s, Wn, Hn)


def cv_mat_to_op_input(img, Wn, Hn):
    """CvMatToOpInput::createArray, one scale: uint8 [B,H0,W0,3] BGR -> float32 [B,3,Hn,Wn] = x/256 - 0.5."""
    r = op_resize_fixed_aspect(img, Wn, Hn).astype(np.float32) / np.float32(256.0) - np.float32(0.5)
    return np.ascontiguousarray(r.transpose(0, 3, 1, 2))
