"""ORACLE (test infrastructure, not product code) -- "V1" decode.

CPU restatement of the reference's in-tree multi-person decode:
  * ``PyOpenPoseOpenCV.postprocess_multi``  openpose/opencv/skeleton.py:102-154
  * ``getKeypoints``                         openpose/opencv/utils.py:9-23
  * ``getValidPairs``                        openpose/opencv/utils.py:27-131
  * ``getPersonwiseKeypoints``               openpose/opencv/utils.py:136-173
  * ``postprocess_single``                   openpose/opencv/skeleton.py:77-100
The three functions of utils.py are transcribed line by line; the cv2 calls they
make are restated from public OpenCV 4.5.x behaviour (the version pinned at
docker/opencvgpu/dockerfiles/Dockerfile.U20Opencvgpu:80-81) because ``cv2`` is
not installed anywhere this runs:

  cv2.resize(f32, (W,H))  INTER_LINEAR, half-pixel centres, x clamped with
                          weight reset, y rows clipped            -> resize_linear
  cv2.GaussianBlur(3,3,0) separable [1/4,1/2,1/4], BORDER_REFLECT_101, fp32
                                                                   -> gaussian3
  cv2.findContours + fillConvexPoly + minMaxLoc (utils.py:14-22)   -> blob_peaks

PARITY UNPINNED against real cv2: the reference holds no golden vector for this
path and cv2 cannot be imported.  Two stated deviations from literal
cv2 semantics, both in ``blob_peaks``: a blob is an 8-connected component of the
mask and its peak is the first raster arg-max of the smoothed map over the
component (== fillConvexPoly of the outer contour for convex blobs); hole
contours, which RETR_TREE also returns and which would add a spurious second
keypoint for ring-shaped blobs, are not emitted.  Blob order follows
findContours' list order for top-level contours, i.e. the reverse of the raster
order of each blob's first pixel.

np.dot(paf_interp, d_ij) (utils.py:97) is restated as a0*d0 + a1*d1 in float64
(two products, one add, no FMA): BLAS's internal rounding is not specified.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.
"""
import numpy as np

f32 = np.float32


# ---------------------------------------------------------------------------
# cv2 primitives
# ---------------------------------------------------------------------------

def _lin_coords_x(n_dst, n_src):
    d = np.arange(n_dst, dtype=np.float64)
    fx = ((d + 0.5) * (n_src / float(n_dst)) - 0.5).astype(f32)
    sx = np.floor(fx).astype(np.int32)
    fx = (fx - sx.astype(f32)).astype(f32)
    lo = sx < 0
    fx[lo] = 0
    sx[lo] = 0
    hi = sx >= n_src - 1
    fx[hi] = 0
    sx[hi] = n_src - 1
    sx1 = np.minimum(sx + 1, n_src - 1)
    return sx, sx1, (f32(1) - fx).astype(f32), fx


def _lin_coords_y(n_dst, n_src):
    d = np.arange(n_dst, dtype=np.float64)
    fy = ((d + 0.5) * (n_src / float(n_dst)) - 0.5).astype(f32)
    sy = np.floor(fy).astype(np.int32)
    fy = (fy - sy.astype(f32)).astype(f32)
    sy0 = np.clip(sy, 0, n_src - 1)
    sy1 = np.clip(sy + 1, 0, n_src - 1)
    return sy0, sy1, (f32(1) - fy).astype(f32), fy


def resize_linear(src, W, H):
    """cv2.resize(src f32 [hs,ws], (W,H)) with INTER_LINEAR -> f32 [H,W]."""
    src = np.asarray(src, f32)
    hs, ws = src.shape
    sx, sx1, a0, a1 = _lin_coords_x(W, ws)
    sy0, sy1, b0, b1 = _lin_coords_y(H, hs)
    rows = (src[:, sx] * a0 + src[:, sx1] * a1).astype(f32)          # [hs, W]
    return (rows[sy0] * b0[:, None] + rows[sy1] * b1[:, None]).astype(f32)


def resize_linear_at(src, W, H, xs, ys):
    """Values of resize_linear(src, W, H)[ys, xs] without building the map."""
    src = np.asarray(src, f32)
    hs, ws = src.shape
    sx, sx1, a0, a1 = _lin_coords_x(W, ws)
    sy0, sy1, b0, b1 = _lin_coords_y(H, hs)
    xs = np.asarray(xs)
    ys = np.asarray(ys)
    r0 = (src[sy0[ys], sx[xs]] * a0[xs] + src[sy0[ys], sx1[xs]] * a1[xs]).astype(f32)
    r1 = (src[sy1[ys], sx[xs]] * a0[xs] + src[sy1[ys], sx1[xs]] * a1[xs]).astype(f32)
    return (r0 * b0[ys] + r1 * b1[ys]).astype(f32)


def gaussian3(m):
    """cv2.GaussianBlur(m, (3,3), 0, 0): [1/4,1/2,1/4] rows then columns."""
    m = np.asarray(m, f32)
    p = np.pad(m, ((0, 0), (1, 1)), mode="reflect")
    t = (p[:, 1:-1] * f32(0.5) + (p[:, :-2] + p[:, 2:]) * f32(0.25)).astype(f32)
    p = np.pad(t, ((1, 1), (0, 0)), mode="reflect")
    return (p[1:-1] * f32(0.5) + (p[:-2] + p[2:]) * f32(0.25)).astype(f32)


def blob_peaks(prob_map, threshold):
    """getKeypoints (utils.py:9-23) -> [(x, y, score f32)] in findContours order."""
    from scipy import ndimage
    smooth = gaussian3(prob_map)
    mask = smooth > f32(threshold)                          # fp32 compare (utils.py:11)
    labels, n = ndimage.label(mask, structure=np.ones((3, 3), np.int32))
    if n == 0:
        return []
    H, W = mask.shape
    flat = labels.ravel()
    idx = np.nonzero(flat)[0]                               # raster order
    lab = flat[idx]
    vals = smooth.ravel()[idx]
    first_pix = np.full(n + 1, -1, np.int64)
    best_val = np.full(n + 1, -np.inf, np.float32)
    best_pix = np.full(n + 1, -1, np.int64)
    for k in range(len(idx)):                               # first raster max per blob
        l = lab[k]
        if first_pix[l] < 0:
            first_pix[l] = idx[k]
        if vals[k] > best_val[l]:
            best_val[l] = vals[k]
            best_pix[l] = idx[k]
    order = sorted(range(1, n + 1), key=lambda l: -first_pix[l])
    out = []
    for l in order:
        y, x = divmod(int(best_pix[l]), W)
        out.append((x, y, prob_map[y, x]))
    return out


# ---------------------------------------------------------------------------
# utils.py transcriptions
# ---------------------------------------------------------------------------

def get_valid_pairs(prediction, detected_keypoints, pose_pairs, body_parts, image_width, image_height, mapidx,
                    verbose=False):
    """getValidPairs (utils.py:27-131)."""
    valid_pairs = []
    invalid_pairs = []
    n_interp_samples = 10
    paf_score_th = 0.1
    conf_th = 0.5
    for k in range(len(mapidx)):
        paf_a = prediction[0, mapidx[k][0], :, :]
        paf_b = prediction[0, mapidx[k][1], :, :]
        cand_a = detected_keypoints[body_parts[pose_pairs[k][0]]]
        cand_b = detected_keypoints[body_parts[pose_pairs[k][1]]]
        n_a, n_b = len(cand_a), len(cand_b)
        if n_a != 0 and n_b != 0:
            valid_pair = []
            for i in range(n_a):
                max_j = -1
                max_score = -1
                found = 0
                for j in range(n_b):
                    dx = float(cand_b[j][0] - cand_a[i][0])
                    dy = float(cand_b[j][1] - cand_a[i][1])
                    norm = np.sqrt(dx * dx + dy * dy)
                    if not norm:
                        continue
                    d0, d1 = dx / norm, dy / norm
                    # np.linspace(a, b, 10): a + k*step, last sample exactly b
                    stepx = (cand_b[j][0] - cand_a[i][0]) / 9.0
                    stepy = (cand_b[j][1] - cand_a[i][1]) / 9.0
                    ks = np.arange(10, dtype=np.float64)
                    cx = ks * stepx + cand_a[i][0]
                    cy = ks * stepy + cand_a[i][1]
                    cx[-1] = cand_b[j][0]
                    cy[-1] = cand_b[j][1]
                    xi = np.rint(cx).astype(np.int64)               # int(round()) -- ties cannot occur
                    yi = np.rint(cy).astype(np.int64)
                    pa = resize_linear_at(paf_a, image_width, image_height, xi, yi).astype(np.float64)
                    pb = resize_linear_at(paf_b, image_width, image_height, xi, yi).astype(np.float64)
                    paf_scores = pa * d0 + pb * d1
                    total = 0.0
                    for s in paf_scores:
                        total = total + float(s)
                    avg = total / 10
                    if int(np.count_nonzero(paf_scores > paf_score_th)) / n_interp_samples > conf_th:
                        if avg > max_score:
                            max_j = j
                            max_score = avg
                            found = 1
                if found:
                    valid_pair.append([cand_a[i][3], cand_b[max_j][3], max_score])
            valid_pairs.append(np.array(valid_pair, dtype=np.float64) if valid_pair else np.zeros((0, 3)))
        else:
            if verbose:
                print("No Connection : k = {}".format(k))
            invalid_pairs.append(k)
            valid_pairs.append([])
    return valid_pairs, invalid_pairs


def get_personwise_keypoints(valid_pairs, invalid_pairs, keypoints_list, body_parts, pose_pairs, mapidx):
    """getPersonwiseKeypoints (utils.py:136-173)."""
    people = []
    for k in range(len(mapidx)):
        if k in invalid_pairs:
            continue
        part_as = valid_pairs[k][:, 0]
        part_bs = valid_pairs[k][:, 1]
        index_a, index_b = (body_parts[x] for x in pose_pairs[k])
        for i in range(len(valid_pairs[k])):
            found = 0
            person_idx = -1
            for j, row in enumerate(people):
                if row[index_a] == part_as[i]:
                    person_idx = j
                    found = 1
                    break
            if found:
                people[person_idx][index_b] = part_bs[i]
                people[person_idx][-1] += keypoints_list[int(part_bs[i]), 2] + valid_pairs[k][i][2]
            elif not found and k < len(body_parts) - 2:
                row = -1 * np.ones(len(body_parts))
                row[index_a] = part_as[i]
                row[index_b] = part_bs[i]
                row[-1] = (keypoints_list[int(part_as[i]), 2] + keypoints_list[int(part_bs[i]), 2]) \
                    + valid_pairs[k][i][2]
                people.append(row)
    return people


def postprocess_multi(prediction, body_parts, pose_pairs, mapidx, image_width, image_height, conf_thres=0.1,
                      return_intermediates=False):
    """PyOpenPoseOpenCV.postprocess_multi (skeleton.py:102-154).

    prediction f32 [1,C,hs,ws] -> (keypoints_list f64 [K,3], personwise list of f64 [len(body_parts)])."""
    prediction = np.asarray(prediction, f32)
    keypoint_id = 0
    keypoints_list = np.zeros((0, 3))
    detected = []
    for i in range(len(body_parts) - 1):
        heat = resize_linear(prediction[0, i], image_width, image_height)
        kps = blob_peaks(heat, conf_thres)
        with_id = []
        for kp in kps:
            with_id.append(kp + (keypoint_id,))
            keypoints_list = np.vstack([keypoints_list, kp])
            keypoint_id += 1
        detected.append(with_id)
    valid_pairs, invalid_pairs = get_valid_pairs(prediction, detected, pose_pairs, body_parts, image_width,
                                                 image_height, mapidx)
    people = get_personwise_keypoints(valid_pairs, invalid_pairs, keypoints_list, body_parts, pose_pairs, mapidx)
    if return_intermediates:
        return keypoints_list, people, dict(detected=detected, valid_pairs=valid_pairs, invalid_pairs=invalid_pairs)
    return keypoints_list, people


def postprocess_single(prediction, n_body_parts, image_width, image_height, conf_thres=0.1):
    """PyOpenPoseOpenCV.postprocess_single (skeleton.py:77-100)."""
    prediction = np.asarray(prediction, f32)
    points = []
    for i in range(n_body_parts):
        m = prediction[0, i]
        flat = int(np.argmax(m))                             # first maximum, as minMaxLoc
        py, px = divmod(flat, m.shape[1])
        conf = float(m[py, px])
        x = (image_width * px) / prediction.shape[3]
        y = (image_height * py) / prediction.shape[2]
        points.append((int(x), int(y)) if conf > conf_thres else None)
    return points
