"""Deterministic synthetic inputs (no datasets or model files exist offline).

``synth_net_output`` draws P stick-figure people and renders what a trained
OpenPose network would emit for them at stride 8: Gaussian-blob part heat maps
plus unit-vector PAF ribbons along each limb.  Used by the
parity tests, the decode-stress benchmark and smoke().
"""
import numpy as np


def synth_people(rng, n_people, n_parts, hs, ws, pairs, jitter=1.5):
    """Random skeletons in net-output cell coordinates -> f32 [P, n_parts, 2],
    bool visibility [P, n_parts]."""
    people = np.zeros((n_people, n_parts, 2), np.float32)
    vis = np.ones((n_people, n_parts), bool)
    for p in range(n_people):
        cx = rng.uniform(4, ws - 4)
        cy = rng.uniform(6, hs - 6)
        size = rng.uniform(4.0, 9.0)
        pts = {pairs[0][0]: np.array([cx, cy])}
        # grow the tree limb by limb with a random direction biased downwards
        for (a, b) in pairs:
            if a in pts and b not in pts:
                ang = rng.uniform(0, 2 * np.pi)
                ln = size * rng.uniform(0.5, 1.0)
                pts[b] = pts[a] + ln * np.array([np.cos(ang), np.sin(ang)])
            elif b in pts and a not in pts:
                ang = rng.uniform(0, 2 * np.pi)
                ln = size * rng.uniform(0.5, 1.0)
                pts[a] = pts[b] + ln * np.array([np.cos(ang), np.sin(ang)])
        for k in range(n_parts):
            q = pts.get(k, np.array([cx, cy])) + rng.normal(0, jitter * 0.1, 2)
            people[p, k] = np.clip(q, [1.0, 1.0], [ws - 2.0, hs - 2.0])
            if rng.uniform() < 0.08:
                vis[p, k] = False
    return people, vis


def synth_net_output(seed, n_people, model, hs=46, ws=82, sigma=0.9, paf_width=0.8, noise=0.003):
    """-> f32 [C, hs, ws] with C = model.n_channels (heat maps incl. background
    first, then PAF x/y channel pairs in model.mapidx order)."""
    rng = np.random.default_rng(seed)
    J, C = model.n_parts, model.n_channels
    out = np.zeros((C, hs, ws), np.float32)
    yy, xx = np.mgrid[0:hs, 0:ws].astype(np.float32)
    people, vis = synth_people(rng, n_people, J, hs, ws, model.pairs)
    for p in range(n_people):
        amp = rng.uniform(0.6, 0.95)
        for k in range(J):
            if not vis[p, k]:
                continue
            x0, y0 = people[p, k]
            g = amp * np.exp(-((xx - x0) ** 2 + (yy - y0) ** 2) / (2 * sigma ** 2))
            out[k] = np.maximum(out[k], g)
        for li, (a, b) in enumerate(model.pairs):
            if not (vis[p, a] and vis[p, b]):
                continue
            pa, pb = people[p, a], people[p, b]
            d = pb - pa
            ln = float(np.hypot(d[0], d[1]))
            if ln < 1e-3:
                continue
            u = d / ln
            # distance of every cell to the segment
            t = np.clip(((xx - pa[0]) * u[0] + (yy - pa[1]) * u[1]) / ln, 0, 1)
            dist = np.hypot(xx - (pa[0] + t * d[0]), yy - (pa[1] + t * d[1]))
            m = (dist <= paf_width).astype(np.float32)
            cx, cy = model.n_heat + model.mapidx[li][0], model.n_heat + model.mapidx[li][1]
            out[cx] = np.where(m > 0, u[0], out[cx])
            out[cy] = np.where(m > 0, u[1], out[cy])
    out[J] = 1.0 - out[:J].max(axis=0)                      # background
    out += rng.normal(0, noise, out.shape).astype(np.float32)
    return out.astype(np.float32)


def synth_frames(seed, batch, height, width):
    """Seeded uniform-u8 BGR frames [B,H,W,3]."""
    rng = np.random.default_rng(seed)
    return rng.integers(0, 256, (batch, height, width, 3), dtype=np.uint8)
