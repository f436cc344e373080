"""ORACLE (test infrastructure, not product code) -- "V2" decode.

CPU restatement, in NumPy float32, of the decode that runs inside
``op.WrapperPython.emplaceAndPop`` on the reference's native path
(``openpose/native/python/skeleton.py:444-450``): net output -> bicubic x8
upsample -> 3x3 NMS with 7x7 sub-pixel refine -> PAF line-integral limb scoring
-> person assembly (two upstream routes: per-limb sorted greedy matching + createPeopleVector in CPU builds;
one globally sorted connection list with assign / extend / merge-if-disjoint in CUDA builds, which is what the
reference's Docker image runs) -> person filter -> ``poseKeypoints [M,J,3]`` / ``poseScores [M]``.

PARITY UNPINNED.  That arithmetic lives in the third-party OpenPose library
(author's fork ``cheneeheng/openpose`` branch ``devel``, no commit pin,
self-identified 1.7.0; ``docker/openpose/dockerfiles/Dockerfile.U20:66-69``),
whose source is not under /root/reference and which cannot be built or
imported here.  This file restates the published upstream 1.7.0 algorithm
(``resizeAndMergeBase.cu`` bicubic, ``nmsBase.cu``, ``bodyPartConnectorBase.cpp``)
from knowledge of that source; the only reference-held artefact for this path,
``data/test/output/inference_native/predictions.txt``, pins the output FORMAT
(score + 25x(x,y,score), zero triplet for a missing joint, score = (sum of part
scores + sum of limb scores)/51) and is checked in tests/test_oracle_v2.py.

Floating-point contract shared with the HIP kernels (csrc/decode_v2.hip): every
operation below is a single IEEE fp32 operation in the written order, no FMA
contraction, so the GPU path (built with -ffp-contract=off) is bit-identical.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.
"""
import numpy as np

f32 = np.float32


# ---------------------------------------------------------------------------
# bicubic upsample (upstream resizeAndMergeBase.cu / cuda.hu bicubicInterpolate)
# ---------------------------------------------------------------------------

def _cubic(v0, v1, v2, v3, d):
    """Catmull-Rom cubic in the exact fp32 operation order of the kernels."""
    a = (((f32(-0.5) * v0 + f32(1.5) * v1) - f32(1.5) * v2) + f32(0.5) * v3)
    b = (((v0 - f32(2.5) * v1) + f32(2.0) * v2) - f32(0.5) * v3)
    c = f32(0.5) * (v0 - v2)
    return (((a * d) * d) * d + (b * d) * d - c * d) + v1


def _src_coords(n_target, n_source):
    """xSource = (x + 0.5) * ws / wt - 0.5 ; 4 clamped taps + fraction."""
    x = np.arange(n_target, dtype=f32)
    xs = ((x + f32(0.5)) * f32(n_source)) / f32(n_target) - f32(0.5)
    x1f = np.floor(xs)
    x1 = np.clip(x1f.astype(np.int32), 0, n_source - 1)
    x0 = np.maximum(0, x1 - 1)
    x2 = np.minimum(n_source - 1, x1 + 1)
    x3 = np.minimum(n_source - 1, x2 + 1)
    dx = (xs - x1.astype(f32)).astype(f32)
    return x0, x1, x2, x3, dx


def upsample_bicubic(src, h_target, w_target):
    """src f32 [..., hs, ws] -> f32 [..., h_target, w_target].

    temp[i] = cubic over x on source row yInt[i]; out = cubic over y of temp --
    identical to upstream's per-pixel bicubicInterpolate because temp[i]
    depends only on (source row, x).
    """
    src = np.asarray(src, dtype=f32)
    hs, ws = src.shape[-2:]
    x0, x1, x2, x3, dx = _src_coords(w_target, ws)
    y0, y1, y2, y3, dy = _src_coords(h_target, hs)
    t = _cubic(src[..., :, x0], src[..., :, x1], src[..., :, x2],
               src[..., :, x3], dx)                       # [..., hs, wt]
    dyc = dy[:, None]
    out = _cubic(t[..., y0, :], t[..., y1, :], t[..., y2, :], t[..., y3, :],
                 dyc)
    return out.astype(f32)


# ---------------------------------------------------------------------------
# NMS (upstream nmsBase.cu: nmsRegisterKernel + scan + writeResultKernel)
# ---------------------------------------------------------------------------

def nms(up, threshold, max_peaks=127, offset=(0.5, 0.5)):
    """up f32 [P, H, W] -> peaks f32 [P, max_peaks+1, 3].

    peaks[p,0,0] = min(count, max_peaks); rows 1.. = (x, y, score) in raster
    (y, x) order of the integer peak location; strict > against threshold and
    all 8 neighbours; 1-px border excluded; 7x7 score-weighted centroid over
    positive scores, + offset.
    """
    up = np.asarray(up, dtype=f32)
    P, H, W = up.shape
    out = np.zeros((P, max_peaks + 1, 3), dtype=f32)
    thr = f32(threshold)
    c = up[:, 1:-1, 1:-1]
    is_peak = c > thr
    for dy in (-1, 0, 1):
        for dx in (-1, 0, 1):
            if dy == 0 and dx == 0:
                continue
            is_peak &= c > up[:, 1 + dy:H - 1 + dy, 1 + dx:W - 1 + dx]
    ox, oy = f32(offset[0]), f32(offset[1])
    for p in range(P):
        ys, xs = np.nonzero(is_peak[p])           # raster order
        ys = ys + 1
        xs = xs + 1
        n = len(ys)
        out[p, 0, 0] = f32(min(n, max_peaks))
        for k in range(min(n, max_peaks)):
            py, px = int(ys[k]), int(xs[k])
            xacc = f32(0.0)
            yacc = f32(0.0)
            sacc = f32(0.0)
            for yy in range(py - 3, py + 4):
                if yy < 0 or yy >= H:
                    continue
                for xx in range(px - 3, px + 4):
                    if xx < 0 or xx >= W:
                        continue
                    s = up[p, yy, xx]
                    if s > 0:
                        xacc = f32(xacc + f32(xx) * s)
                        yacc = f32(yacc + f32(yy) * s)
                        sacc = f32(sacc + s)
            out[p, k + 1, 0] = f32(xacc / sacc) + ox
            out[p, k + 1, 1] = f32(yacc / sacc) + oy
            out[p, k + 1, 2] = up[p, py, px]
    return out


# ---------------------------------------------------------------------------
# PAF scoring (upstream bodyPartConnectorBase.cpp getScoreAB / .cu process)
# ---------------------------------------------------------------------------

def _pir(x):
    """positiveIntRound: int(x + 0.5f) on fp32."""
    return int(f32(f32(x) + f32(0.5)))


def _close_thr(W, H, cuda):
    """Distance below which two unconnected peaks still get the minimum PAF score: sqrt(W*H)/150, evaluated in
    double by the CPU connector (bodyPartConnectorBase.cpp getScoreAB) and in float by the CUDA one
    (bodyPartConnectorBase.cu process(): sqrtf(w*h)/150, compared with the float distance)."""
    if cuda:
        return float(f32(np.sqrt(f32(W * H))) / f32(150.0))
    return np.sqrt(float(W * H)) / 150.0


def score_ab(a, b, map_x, map_y, inter_threshold, inter_min_above, nms_thr, cuda=False):
    """a, b = (x, y, score) fp32 rows; map_x/map_y f32 [H, W] upsampled PAFs.

    Returns fp32 score (0 when rejected), as upstream getScoreAB (cuda=True: the CUDA build's process(), which
    differs only in the precision of the close-keypoint threshold and returns -1 instead of 0 when rejected --
    both are discarded by the `> 1e-6` test of their callers).
    """
    H, W = map_x.shape
    vx = f32(b[0] - a[0])
    vy = f32(b[1] - a[1])
    vmax = max(abs(vx), abs(vy))
    npts = max(5, min(25, _pir(np.sqrt(f32(f32(5.0) * vmax)))))
    norm = np.sqrt(f32(f32(vx * vx) + f32(vy * vy)))
    if float(norm) > 1e-6:
        sx, sy = f32(a[0]), f32(a[1])
        nx = f32(vx / norm)
        ny = f32(vy / norm)
        total = f32(0.0)
        count = 0
        stepx = f32(vx / f32(npts))
        stepy = f32(vy / f32(npts))
        for lm in range(npts):
            mx = max(0, min(W - 1, _pir(f32(sx + f32(f32(lm) * stepx)))))
            my = max(0, min(H - 1, _pir(f32(sy + f32(f32(lm) * stepy)))))
            s = f32(f32(nx * map_x[my, mx]) + f32(ny * map_y[my, mx]))
            if s > f32(inter_threshold):
                total = f32(total + s)
                count += 1
        if f32(f32(count) / f32(npts)) > f32(inter_min_above):
            return f32(total / f32(count))
        # close-keypoint rescue (1.7.0): minimum PAF score for near-coincident
        l2 = norm
        thr = _close_thr(W, H, cuda)
        if float(l2) < thr:
            return f32(float(f32(nms_thr)) + 1e-6)
    return f32(0.0)


def pair_scores(cand_a, cand_b, map_x, map_y, inter_threshold, inter_min_above, nms_thr, cuda=False):
    """Vectorised score_ab over all (i, j): cand_a f32 [nA,3], cand_b f32 [nB,3]
    -> f32 [nA, nB].  Same fp32 operations, element-wise, as score_ab."""
    H, W = map_x.shape
    ax, ay = cand_a[:, 0:1].astype(f32), cand_a[:, 1:2].astype(f32)
    bx, by = cand_b[None, :, 0].astype(f32), cand_b[None, :, 1].astype(f32)
    vx = (bx - ax).astype(f32)
    vy = (by - ay).astype(f32)
    vmax = np.maximum(np.abs(vx), np.abs(vy))
    npts = (np.sqrt(f32(5.0) * vmax) + f32(0.5)).astype(np.int32)
    npts = np.maximum(5, np.minimum(25, npts))
    norm = np.sqrt(vx * vx + vy * vy).astype(f32)
    ok = norm.astype(np.float64) > 1e-6
    safe = np.where(ok, norm, f32(1.0))
    nx = (vx / safe).astype(f32)
    ny = (vy / safe).astype(f32)
    nf = npts.astype(f32)
    stepx = (vx / nf).astype(f32)
    stepy = (vy / nf).astype(f32)
    total = np.zeros_like(vx)
    count = np.zeros(vx.shape, np.int32)
    sx = np.broadcast_to(ax, vx.shape)
    sy = np.broadcast_to(ay, vx.shape)
    for lm in range(25):
        act = ok & (lm < npts)
        if not act.any():
            break
        mx = ((sx + f32(lm) * stepx) + f32(0.5)).astype(np.int32)
        my = ((sy + f32(lm) * stepy) + f32(0.5)).astype(np.int32)
        mx = np.clip(mx, 0, W - 1)
        my = np.clip(my, 0, H - 1)
        s = (nx * map_x[my, mx] + ny * map_y[my, mx]).astype(f32)
        hit = act & (s > f32(inter_threshold))
        total = np.where(hit, total + s, total).astype(f32)
        count += hit
    cf = count.astype(f32)
    good = ok & ((cf / nf).astype(f32) > f32(inter_min_above))
    with np.errstate(divide="ignore", invalid="ignore"):
        avg = (total / cf).astype(f32)
    close = ok & ~good & (norm.astype(np.float64) < _close_thr(W, H, cuda))
    out = np.zeros_like(vx)
    out = np.where(good, avg, out)
    out = np.where(close, f32(float(f32(nms_thr)) + 1e-6), out)
    return out.astype(f32)


# ---------------------------------------------------------------------------
# person assembly (upstream createPeopleVector / removePeopleBelowThresholds /
# peopleVectorToPeopleArray)
# ---------------------------------------------------------------------------

def select_connections(peaks, maps, model, pair_index):
    """All accepted (score, i, j) for one limb, sorted descending by
    (score, i, j) as std::greater<tuple<double,int,int>>, then the greedy
    unique selection.  Returns list of (indexA, indexB, score_f32)."""
    max_peaks = model.max_peaks
    off = 3 * (max_peaks + 1)
    part_a, part_b = model.pairs[pair_index]
    na = _pir(peaks[part_a, 0, 0])
    nb = _pir(peaks[part_b, 0, 0])
    if na == 0 or nb == 0:
        return []
    ch_x = model.n_heat + model.mapidx[pair_index][0]
    ch_y = model.n_heat + model.mapidx[pair_index][1]
    sc = pair_scores(peaks[part_a, 1:na + 1], peaks[part_b, 1:nb + 1], maps[ch_x], maps[ch_y],
                     model.inter_threshold, model.inter_min_above_threshold, model.nms_threshold)
    ii, jj = np.nonzero(sc.astype(np.float64) > 1e-6)
    cands = [(float(sc[i, j]), int(i) + 1, int(j) + 1) for i, j in zip(ii, jj)]
    cands.sort(reverse=True)
    conns = []
    min_ab = min(na, nb)
    occ_a = [0] * na
    occ_b = [0] * nb
    counter = 0
    for s, i, j in cands:
        if not occ_a[i - 1] and not occ_b[j - 1]:
            conns.append((part_a * off + i * 3 + 2, part_b * off + j * 3 + 2,
                          f32(s)))
            counter += 1
            if counter == min_ab:
                break
            occ_a[i - 1] = 1
            occ_b[j - 1] = 1
    return conns


def assemble(peaks, connections, model):
    """createPeopleVector given per-limb connection lists.

    Returns (rows int32 [R, n_parts+1] (last = part count), scores f32 [R]).
    """
    n_parts = model.n_parts
    off = 3 * (model.max_peaks + 1)
    flat = peaks.reshape(-1)
    rows = []
    scores = []
    for pair_index, (part_a, part_b) in enumerate(model.pairs):
        na = _pir(peaks[part_a, 0, 0])
        nb = _pir(peaks[part_b, 0, 0])
        if na == 0 or nb == 0:
            # add the non-empty side's unclaimed peaks as single-part people
            # (non-MPI branch; numberBodyParts != 15)
            if na == 0:
                part, n = part_b, nb
            else:
                part, n = part_a, na
            for i in range(1, n + 1):
                idx = part * off + i * 3 + 2
                if not any(r[part] == idx for r in rows):
                    r = [0] * (n_parts + 1)
                    r[part] = idx
                    r[-1] = 1
                    rows.append(r)
                    scores.append(f32(flat[idx]))
            continue
        conns = connections[pair_index]
        if not conns:
            continue
        if pair_index == 0:
            for ia, ib, s in conns:
                r = [0] * (n_parts + 1)
                r[part_a] = ia
                r[part_b] = ib
                r[-1] = 2
                rows.append(r)
                scores.append(f32(f32(flat[ia] + flat[ib]) + s))
        elif pair_index in model.ear_pair_indices:
            for ia, ib, s in conns:
                for r in rows:
                    if r[part_a] == ia and r[part_b] == 0:
                        r[part_b] = ib
                    elif r[part_b] == ib and r[part_a] == 0:
                        r[part_a] = ia
        else:
            for ia, ib, s in conns:
                found = False
                for k, r in enumerate(rows):
                    if r[part_a] == ia:
                        r[part_b] = ib
                        r[-1] += 1
                        scores[k] = f32(scores[k] + f32(flat[ib] + s))
                        found = True
                        break
                if not found:
                    r = [0] * (n_parts + 1)
                    r[part_a] = ia
                    r[part_b] = ib
                    r[-1] = 2
                    rows.append(r)
                    scores.append(f32(f32(flat[ia] + flat[ib]) + s))
    rows = np.asarray(rows, dtype=np.int32).reshape(-1, n_parts + 1)
    return rows, np.asarray(scores, dtype=f32)


# ---------------------------------------------------------------------------
# The CUDA build's connector (upstream bodyPartConnectorBase.cu connectBodyPartsGpu ->
# bodyPartConnectorBase.cpp pafPtrIntoVector + pafVectorIntoPeopleVector).  This is the route the reference's
# Docker image runs (docker/openpose/dockerfiles/Dockerfile.U20:66-95 builds OpenPose with CUDA), reached from
# openpose/native/python/skeleton.py:447.  createPeopleVector above is what a CPU-only / OpenCL build runs.
# ---------------------------------------------------------------------------

def paf_ptr_into_vector(peaks, maps, model):
    """pafPtrIntoVector: every (limb, A_i, B_j) whose PAF score passes `> 1e-6`, as
    (totalScore, pafScore, pairIndex, indexA, indexB) with totalScore = paf + 0.1 * scoreA + 0.1 * scoreB (fp32,
    left to right), sorted descending as std::greater<std::tuple<double, double, int, int, int>>."""
    conns = []
    for k, (part_a, part_b) in enumerate(model.pairs):
        na = _pir(peaks[part_a, 0, 0])
        nb = _pir(peaks[part_b, 0, 0])
        if na == 0 or nb == 0:
            continue
        ch_x = model.n_heat + model.mapidx[k][0]
        ch_y = model.n_heat + model.mapidx[k][1]
        sc = pair_scores(peaks[part_a, 1:na + 1], peaks[part_b, 1:nb + 1], maps[ch_x], maps[ch_y],
                         model.inter_threshold, model.inter_min_above_threshold, model.nms_threshold, cuda=True)
        ii, jj = np.nonzero(sc.astype(np.float64) > 1e-6)
        for i, j in zip(ii, jj):
            paf = f32(sc[i, j])
            total = f32(f32(paf + f32(f32(0.1) * peaks[part_a, i + 1, 2])) + f32(f32(0.1) * peaks[part_b, j + 1, 2]))
            conns.append((float(total), float(paf), k, int(i) + 1, int(j) + 1))
    conns.sort(reverse=True)
    return conns


def paf_vector_into_people_vector(conns, peaks, model):
    """pafVectorIntoPeopleVector: walk the globally sorted connections once; per connection
      1. neither end assigned     -> new person {A, B}
      2./3. one end assigned      -> add the other end to that person unless it already has that part
      4. both ends, same person   -> add the PAF score to the person (redundant limb of a cycle)
      5. both ends, two people    -> merge them when they share no part (the lower index survives)
    People emptied by a merge are removed at the end; order = creation order.
    Returns (rows int32 [R, n_parts+1], scores f32 [R])."""
    n_parts, mp = model.n_parts, model.max_peaks
    off = 3 * (mp + 1)
    flat = peaks.reshape(-1)
    rows, scores = [], []
    assigned = np.full((n_parts * mp,), -1, np.int64)
    for _total, paf, k, ia, ib in conns:
        paf = f32(paf)
        part_a, part_b = model.pairs[k]
        idx_a = part_a * off + ia * 3 + 2
        idx_b = part_b * off + ib * 3 + 2
        sa, sb = part_a * mp + ia - 1, part_b * mp + ib - 1
        pa, pb = int(assigned[sa]), int(assigned[sb])
        if pa < 0 and pb < 0:
            r = [0] * (n_parts + 1)
            r[part_a] = idx_a
            r[part_b] = idx_b
            r[-1] = 2
            assigned[sa] = assigned[sb] = len(rows)
            rows.append(r)
            scores.append(f32(f32(flat[idx_a] + flat[idx_b]) + paf))
        elif (pa >= 0) != (pb >= 0):
            p1 = pa if pa >= 0 else pb
            part2, idx2, slot2 = (part_b, idx_b, sb) if pa >= 0 else (part_a, idx_a, sa)
            if rows[p1][part2] == 0:
                rows[p1][part2] = idx2
                rows[p1][-1] += 1
                scores[p1] = f32(scores[p1] + f32(flat[idx2] + paf))
                assigned[slot2] = p1
        elif pa == pb:
            scores[pa] = f32(scores[pa] + paf)
        else:
            p1, p2 = min(pa, pb), max(pa, pb)
            if not any(rows[p1][q] > 0 and rows[p2][q] > 0 for q in range(n_parts)):
                for q in range(n_parts):
                    if rows[p1][q] == 0:
                        rows[p1][q] = rows[p2][q]
                rows[p1][-1] += rows[p2][-1]
                scores[p1] = f32(scores[p1] + f32(scores[p2] + paf))
                rows[p2][-1] = 0
                scores[p2] = f32(0.0)
                assigned[assigned == p2] = p1
    keep = [p for p in range(len(rows)) if rows[p][-1] != 0]
    rows = np.asarray([rows[p] for p in keep], dtype=np.int32).reshape(-1, n_parts + 1)
    return rows, np.asarray([scores[p] for p in keep], dtype=f32)


def filter_people(rows, scores, model, maximize_positives=False):
    """removePeopleBelowThresholdsAndFillFaces -> valid row indices."""
    valid = []
    for p in range(len(rows)):
        counter = int(rows[p, -1])
        if not maximize_positives and model.n_parts == 25:
            feet = int(np.count_nonzero(rows[p, 19:25] > 0))
            if feet > 0:
                counter += 0 - feet
        if counter >= model.min_subset_cnt and \
                f32(scores[p] / f32(counter)) >= f32(model.min_subset_score):
            valid.append(p)
    if not valid and not maximize_positives:
        return filter_people(rows, scores, model, True)
    return valid


def people_array(rows, scores, valid, peaks, model, scale=1.0):
    """peopleVectorToPeopleArray -> (keypoints f32 [M,J,3], scores f32 [M])."""
    flat = peaks.reshape(-1)
    M = len(valid)
    kp = np.zeros((M, model.n_parts, 3), dtype=f32)
    sc = np.zeros((M,), dtype=f32)
    denom = f32(model.n_parts + model.n_pairs)
    for m, p in enumerate(valid):
        for part in range(model.n_parts):
            idx = int(rows[p, part])
            if idx > 0:
                kp[m, part, 0] = f32(flat[idx - 2] * f32(scale))
                kp[m, part, 1] = f32(flat[idx - 1] * f32(scale))
                kp[m, part, 2] = flat[idx]
        sc[m] = f32(scores[p] / denom)
    return kp, sc


def scale_net_to_output(image_w, image_h, net_w, net_h):
    """mScaleNetToOutput of poseExtractorCaffe.cpp::forwardPass (1.7.0), float32: the factor that takes peak
    coordinates on the net-input-sized maps back to input-image pixels.  s = resizeGetScaleFactor(image, net);
    netSize = positiveIntRound(s * image); result = float(resizeGetScaleFactor(netSize, image))."""
    s = min((net_w - 1) / float(image_w - 1), (net_h - 1) / float(image_h - 1))
    nw, nh = int(s * image_w + 0.5), int(s * image_h + 0.5)
    return f32(min((image_w - 1) / float(nw - 1), (image_h - 1) / float(nh - 1)))


def decode(net_out, model, h_target=None, w_target=None, scale=1.0,
           return_intermediates=False, connector=None):
    """Full V2 decode of one frame.  net_out f32 [C, hs, ws]; scale = mScaleNetToOutput.
    connector: 0 = createPeopleVector (CPU / OpenCL builds of OpenPose), 1 = pafPtrIntoVector +
    pafVectorIntoPeopleVector (CUDA builds: what the reference's Docker image runs); None = model.connector."""
    net_out = np.asarray(net_out, dtype=f32)
    C, hs, ws = net_out.shape
    h_target = h_target or hs * 8
    w_target = w_target or ws * 8
    up = upsample_bicubic(net_out, h_target, w_target)   # upstream materialises all channels
    # poseExtractorCaffe.cpp (1.7.0): nmsOffset = float(0.5 / double(mScaleNetToOutput)) -- the half-pixel centre
    # in units of the OUTPUT image, since the keypoints are multiplied by the scale afterwards
    off = f32(0.5 / float(f32(scale)))
    peaks = nms(up[:model.n_parts], model.nms_threshold, model.max_peaks, offset=(off, off))
    maps = up
    if connector is None:
        connector = getattr(model, "connector", 0)
    if connector == 1:
        conns = paf_ptr_into_vector(peaks, maps, model)
        rows, scores = paf_vector_into_people_vector(conns, peaks, model)
    else:
        conns = [select_connections(peaks, maps, model, k)
                 for k in range(model.n_pairs)]
        rows, scores = assemble(peaks, conns, model)
    valid = filter_people(rows, scores, model)
    kp, sc = people_array(rows, scores, valid, peaks, model, scale)
    if return_intermediates:
        return kp, sc, dict(peaks=peaks, connections=conns, rows=rows,
                            row_scores=scores, valid=valid)
    return kp, sc
