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
path and cv2 cannot be imported.  ``blob_peaks`` is getKeypoints
literally: Suzuki-Abe border following as OpenCV 4.5 contours.cpp codes it (outer
AND hole borders, RETR_TREE list order = latest sibling first, node before its
children), the CHAIN_APPROX_SIMPLE vertices, FillConvexPoly's outline + two-walker
scan fill, first raster maximum of mapSmooth * mask.  The GPU kernels implement
these literal semantics too (decode_v1.hip, K-B) and are tested against ``blob_peaks``.
``blob_peaks_cc`` is the connected-component short form (8-connected component,
first raster arg-max, reverse raster order of first pixels): equal to the literal
form for every blob with exactly one mask run per row -- the case the GPU kernel
short-cuts -- and different for ring blobs (extra hole keypoint) and for shapes
fillConvexPoly under-fills; known answers in tests/test_oracle_v1.py.

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


def blob_peaks_cc(prob_map, threshold):
    """getKeypoints (utils.py:9-23) in connected-component form: a blob = an 8-connected component of the mask,
    its keypoint = the first raster arg-max of the smoothed map over the component, blobs in reverse raster order
    of their first pixel.  Equal to the literal form (blob_peaks) whenever every blob has exactly one mask run per
    row (then it has no hole and fillConvexPoly paints exactly its pixels) -- the short cut the GPU kernel takes
    for such maps; see tests/test_oracle_v1.py for the ring / L-shape / merged-people cases where the two differ."""
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


# ---- cv2.findContours(mask, RETR_TREE, CHAIN_APPROX_SIMPLE), restated from OpenCV 4.5 contours.cpp -------------
# (cvFindNextContour_Impl / icvFetchContour: Suzuki-Abe border following on a 0/1 image with a zero frame.)
# Direction codes: 0 E, 1 NE, 2 N, 3 NW, 4 W, 5 SW, 6 S, 7 SE (y grows downwards).
_CODE_DX = (1, 1, 0, -1, -1, -1, 0, 1)
_CODE_DY = (0, -1, -1, -1, 0, 1, 1, 1)


def _fetch_contour(img, x0, y0, is_hole, nbd):
    """icvFetchContour / icvFetchContourEx: follow the border that starts at (x0, y0) of the padded int image,
    mark it (nbd, or -nbd where the pixel to the right was examined and found empty) and return the vertices
    CHAIN_APPROX_SIMPLE keeps (a point is written whenever the step direction changes)."""
    pts = []
    s_end = s = 0 if is_hole else 4
    while True:
        s = (s - 1) & 7
        i1x, i1y = x0 + _CODE_DX[s], y0 + _CODE_DY[s]
        if img[i1y, i1x] != 0 or s == s_end:
            break
    if s == s_end:                                   # single pixel domain
        img[y0, x0] = -nbd
        pts.append((x0, y0))
        return pts
    i3x, i3y = x0, y0
    ptx, pty = x0, y0
    prev_s = s ^ 4
    while True:
        s_end = s
        while s < 15:
            s += 1
            i4x, i4y = i3x + _CODE_DX[s & 7], i3y + _CODE_DY[s & 7]
            if img[i4y, i4x] != 0:
                break
        s &= 7
        if ((s - 1) & 0xffffffff) < (s_end & 0xffffffff):          # (unsigned)(s - 1) < (unsigned)s_end
            img[i3y, i3x] = -nbd
        elif img[i3y, i3x] == 1:
            img[i3y, i3x] = nbd
        if s != prev_s:
            pts.append((ptx, pty))
            prev_s = s
        ptx += _CODE_DX[s]
        pty += _CODE_DY[s]
        if (i4x, i4y) == (x0, y0) and (i3x, i3y) == (i1x, i1y):
            break
        i3x, i3y = i4x, i4y
        s = (s + 4) & 7
    return pts


def find_contours_tree(mask):
    """-> list of int arrays [n, 2] (x, y), in the order cv2.findContours(mask, RETR_TREE, CHAIN_APPROX_SIMPLE)
    returns them: outer and hole borders; a border found later is linked in FRONT of its siblings
    (cvInsertNodeIntoTree), and cvTreeToNodeSeq walks the tree node, children, next sibling."""
    H, W = mask.shape
    img = np.zeros((H + 2, W + 2), np.int64)          # copyMakeBorder(1, 1, 1, 1, BORDER_CONSTANT, 0)
    img[1:-1, 1:-1] = (mask != 0)
    contours = []                                      # (points, is_hole, parent) in discovery order; parent -1 = frame
    nbd = 1
    for y in range(1, H + 1):
        lnbd = 0                                       # id of the last border met on this row (0 = the frame)
        row = img[y]
        # only positions where the value changes can start a border; visit them left to right
        x = 1
        prev = 0
        while x <= W + 1:
            p = row[x]
            if p == prev:
                x += 1
                continue
            is_hole = None
            if prev == 0 and p == 1:
                is_hole = False
            elif p == 0 and prev >= 1:
                is_hole = True
            if is_hole is not None:
                if is_hole and prev > 1:
                    lnbd = prev
                # parent: the last border met, or that border's parent when it is of the same kind
                if lnbd == 0:
                    parent = -1
                else:
                    b = lnbd - 2
                    parent = contours[b][2] if contours[b][1] == is_hole else b
                nbd += 1
                sx = x - 1 if is_hole else x
                pts = _fetch_contour(img, sx, y, is_hole, nbd)
                contours.append((np.asarray(pts, np.int64) - 1, is_hole, parent))
                lnbd = nbd
                p = row[x]
            prev = p
            if abs(prev) > 1:
                lnbd = abs(prev)
            x += 1
    children = {}
    for i, (_, _, parent) in enumerate(contours):
        children.setdefault(parent, []).insert(0, i)   # latest first
    out = []

    def walk(node):
        for c in children.get(node, []):
            out.append(contours[c][0])
            walk(c)
    walk(-1)
    return out


def _trunc_div(a, b):
    q = abs(a) // abs(b)
    return q if (a >= 0) == (b > 0) else -q


def fill_convex_poly(shape, pts):
    """cv2.fillConvexPoly(zeros(shape), pts, 1) with the default lineType 8, shift 0 -- OpenCV 4.5 drawing.cpp
    FillConvexPoly restated: the outline is drawn edge by edge (8-connected lines; the edges of a
    CHAIN_APPROX_SIMPLE contour are straight runs), then two edge walkers start at the first vertex of minimal y,
    one forwards and one backwards through the vertex list, and every row gets the span between them.  For a polygon
    that is not convex the walkers still only follow those two chains: the span can miss parts of the shape."""
    H, W = shape
    out = np.zeros(shape, bool)
    n = len(pts)
    v = [(int(p[0]), int(p[1])) for p in pts]
    XY_SHIFT, XY_ONE = 16, 1 << 16
    # outline
    x0, y0 = v[-1]
    for (x1, y1) in v:
        steps = max(abs(x1 - x0), abs(y1 - y0))
        if steps == 0:
            if 0 <= y0 < H and 0 <= x0 < W:
                out[y0, x0] = True
        else:
            dx, dy = (x1 - x0) // steps if (x1 - x0) % steps == 0 else None, (y1 - y0) // steps if (y1 - y0) % steps == 0 else None
            if dx is None or dy is None:
                raise ValueError("fill_convex_poly: an edge that is not a straight 8-connected run")
            for t in range(steps + 1):
                xx, yy = x0 + t * dx, y0 + t * dy
                if 0 <= yy < H and 0 <= xx < W:
                    out[yy, xx] = True
        x0, y0 = x1, y1
    xmin = min(p[0] for p in v)
    xmax = max(p[0] for p in v)
    ymin = min(p[1] for p in v)
    ymax = max(p[1] for p in v)
    imin = 0
    for i, p in enumerate(v):
        if p[1] < v[imin][1]:
            imin = i
    if n < 3 or xmax < 0 or ymax < 0 or xmin >= W or ymin >= H:
        return out
    ymax = min(ymax, H - 1)
    edges = n
    e_idx = [imin, imin]
    e_ye = [ymin, ymin]
    e_di = [1, n - 1]
    e_x = [-XY_ONE, -XY_ONE]
    e_dx = [0, 0]
    delta1 = delta2 = XY_ONE >> 1
    y = ymin
    while True:
        for i in range(2):
            if y >= e_ye[i]:
                idx0, di = e_idx[i], e_di[i]
                idx = idx0 + di
                if idx >= n:
                    idx -= n
                while True:
                    edges -= 1
                    if not edges + 1 > 0:
                        break
                    ty = v[idx][1]
                    if ty > y:
                        xs = v[idx0][0] << XY_SHIFT
                        xe = v[idx][0] << XY_SHIFT
                        e_ye[i] = ty
                        e_dx[i] = _trunc_div((xe - xs) * 2 + (ty - y), 2 * (ty - y))
                        e_x[i] = xs
                        e_idx[i] = idx
                        break
                    idx0 = idx
                    idx += di
                    if idx >= n:
                        idx -= n
        if edges < 0:
            break
        if y >= 0:
            left, right = (1, 0) if e_x[0] > e_x[1] else (0, 1)
            xx1 = (e_x[left] + delta1) >> XY_SHIFT
            xx2 = (e_x[right] + delta2) >> XY_SHIFT
            if xx2 >= 0 and xx1 < W:
                out[y, max(xx1, 0):min(xx2, W - 1) + 1] = True
        e_x[0] += e_dx[0]
        e_x[1] += e_dx[1]
        y += 1
        if not y <= ymax:
            break
    return out


def blob_peaks(prob_map, threshold):
    """getKeypoints (utils.py:9-23), literally: GaussianBlur -> mask -> findContours(RETR_TREE,
    CHAIN_APPROX_SIMPLE) -> per contour (outer AND hole borders, in the list order of findContours):
    fillConvexPoly mask, minMaxLoc of mapSmooth * mask (first raster maximum), score from the unsmoothed map.
    -> [(x, y, score f32)]"""
    smooth = gaussian3(prob_map)
    mask = smooth > f32(threshold)                          # fp32 compare (utils.py:11)
    out = []
    H, W = mask.shape
    for cnt in find_contours_tree(mask):
        x0, y0 = max(int(cnt[:, 0].min()), 0), max(int(cnt[:, 1].min()), 0)
        x1, y1 = min(int(cnt[:, 0].max()), W - 1), min(int(cnt[:, 1].max()), H - 1)
        # the painted pixels lie inside the contour's bounding box: work on that window only
        win = fill_convex_poly((y1 - y0 + 1, x1 - x0 + 1), cnt - np.array([x0, y0]))
        masked = np.where(win, smooth[y0:y1 + 1, x0:x1 + 1].astype(np.float64), 0.0)
        best = float(masked.max())
        if best > 0.0:
            yy, xx = np.unravel_index(int(np.argmax(masked)), masked.shape)       # first raster maximum
            x, y = int(xx) + x0, int(yy) + y0
        else:
            x, y = 0, 0            # minMaxLoc of a map whose maximum is the 0 outside the polygon: its first pixel
            if best < 0.0 or not (masked == 0.0).any():
                raise NotImplementedError("blob_peaks: maximum below zero inside the polygon")
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
                      return_intermediates=False, blobs="opencv"):
    """PyOpenPoseOpenCV.postprocess_multi (skeleton.py:102-154).

    prediction f32 [1,C,hs,ws] -> (keypoints_list f64 [K,3], personwise list of f64 [len(body_parts)]).
    blobs: "opencv" = getKeypoints literally (outer + hole contours, fillConvexPoly); "cc" = the connected-component
    short form (see blob_peaks_cc for when the two agree), kept to show that a scene exercises the difference."""
    prediction = np.asarray(prediction, f32)
    keypoint_id = 0
    keypoints_list = np.zeros((0, 3))
    detected = []
    for i in range(len(body_parts) - 1):
        heat = resize_linear(prediction[0, i], image_width, image_height)
        kps = blob_peaks(heat, conf_thres) if blobs == "opencv" else blob_peaks_cc(heat, conf_thres)
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
