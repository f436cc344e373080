"""GPU parity: fused upsample+NMS / PAF matching / assembly kernels vs the V2
oracle (oracle/decode_v2.py), through the C ABI.  Integer outputs (peak counts,
subset indices) and the fp32 values are bit-exact by construction (same IEEE
operation order, -ffp-contract=off)."""
import numpy as np
import pytest

from posehip import _ffi, synth
from posehip.tables import BODY_25, COCO_18

pytestmark = pytest.mark.gpu


def _oracle(net_out, model, Ht, Wt, scale):
    from oracle import decode_v2
    return decode_v2.decode(net_out, model, Ht, Wt, scale, return_intermediates=True)


def _check_frame(kp_o, sc_o, inter, got, extras, b, model, engine=None):
    kp_g, sc_g = got[b]
    np.testing.assert_array_equal(extras["peaks"][b], inter["peaks"])
    if engine is not None:
        n, idx, cs = engine.decode_v2_connections(model, len(got))
        for k, conns in enumerate(inter["connections"]):
            assert n[b, k] == len(conns), (k, n[b, k], len(conns))
            for c, (ia, ib, s) in enumerate(conns):
                assert (idx[b, k, c, 0], idx[b, k, c, 1]) == (ia, ib)
                assert cs[b, k, c].view(np.uint32) == np.float32(s).view(np.uint32), (k, c, cs[b, k, c], s)
    assert int(extras["n_people"][b]) == len(inter["valid"])
    rows = inter["rows"][inter["valid"]] if len(inter["valid"]) else np.zeros((0, model.n_parts + 1), np.int32)
    np.testing.assert_array_equal(extras["subsets"][b, :len(rows)], rows)
    np.testing.assert_array_equal(kp_g, kp_o)
    np.testing.assert_array_equal(sc_g, sc_o)


@pytest.mark.parametrize("model,hs,ws,people", [
    (BODY_25, 46, 82, [0, 1, 2, 5]),
    (BODY_25, 46, 46, [3]),
    (COCO_18, 46, 82, [4]),
    (BODY_25, 23, 31, [2]),
])
def test_decode_matches_oracle(engine, model, hs, ws, people):
    frames = np.stack([synth.synth_net_output(100 + i, p, model, hs, ws) for i, p in enumerate(people)])
    engine.set_net_out(frames)
    Ht, Wt = hs * 8, ws * 8
    got, extras = engine.decode_v2(model, len(people), Ht, Wt, scale=1.75, want_peaks=True, want_subsets=True)
    found = 0
    for b in range(len(people)):
        kp_o, sc_o, inter = _oracle(frames[b], model, Ht, Wt, 1.75)
        _check_frame(kp_o, sc_o, inter, got, extras, b, model, engine)
        found += len(sc_o)
    if sum(people) > 0:
        assert found > 0                         # the scenes do produce skeletons


def test_crowded_and_truncation(engine):
    # 20 people: stresses ordered compaction; plus a noise frame that overflows 127 peaks/part
    rng = np.random.default_rng(5)
    crowded = synth.synth_net_output(7, 20, BODY_25, 46, 82)
    noise = rng.uniform(0.0, 1.0, crowded.shape).astype(np.float32)
    noise[BODY_25.n_heat:] = rng.normal(0, 0.3, noise[BODY_25.n_heat:].shape)
    frames = np.stack([crowded, noise])
    engine.set_net_out(frames)
    got, extras = engine.decode_v2(BODY_25, 2, 368, 656, want_peaks=True, want_subsets=True, max_people=512)
    assert (extras["peaks"][1, :, 0, 0] == 127).all()       # truncated to max_peaks in raster order
    for b in range(2):
        kp_o, sc_o, inter = _oracle(frames[b], BODY_25, 368, 656, 1.0)
        _check_frame(kp_o, sc_o, inter, got, extras, b, BODY_25, engine)


def test_empty_and_flat_maps(engine):
    z = np.zeros((2, BODY_25.n_channels, 46, 82), np.float32)
    z[1] = 0.5                                   # plateau: strict > never fires
    engine.set_net_out(z)
    got, extras = engine.decode_v2(BODY_25, 2, 368, 656, want_peaks=True)
    assert (extras["n_people"] == 0).all()
    assert (extras["peaks"] == 0).all()
    assert got[0][0].shape == (0, 25, 3) and got[0][1].shape == (0,)


def test_non_x8_target_and_errors(engine):
    from posehip._ffi import PosehipError
    f = synth.synth_net_output(3, 2, BODY_25, 46, 82)[None]
    engine.set_net_out(f)
    got, extras = engine.decode_v2(BODY_25, 1, 46 * 6, 82 * 5, want_peaks=True, want_subsets=True)
    kp_o, sc_o, inter = _oracle(f[0], BODY_25, 46 * 6, 82 * 5, 1.0)
    _check_frame(kp_o, sc_o, inter, got, extras, 0, BODY_25, engine)
    with pytest.raises(PosehipError):
        engine.decode_v2(BODY_25, 1, 46, 82)     # below the supported upsampling ratio
    engine.set_net_out(np.zeros((1, 30, 46, 82), np.float32))
    with pytest.raises(PosehipError):
        engine.decode_v2(BODY_25, 1, 368, 656)   # too few channels for BODY_25


def test_pose_post_matches_oracle_bit_exact(engine):
    """torso boxes (fp32) and depth lift (fp64) computed on the device keypoints of the last decode
    equal oracle/post_v2.py bit for bit, for host and device depth images."""
    import torch
    from oracle import post_v2
    from posehip import synth
    from posehip.tables import BODY_25, TORSO_IDS
    B, hs, ws = 3, 46, 82
    net = np.stack([synth.synth_net_output(900 + i, [4, 0, 7][i], BODY_25, hs, ws) for i in range(B)])
    engine.set_net_out(net)
    res, ex = engine.decode_v2(BODY_25, B, hs * 8, ws * 8, 1080 / 368, 128)
    rng = np.random.default_rng(5)
    depth = rng.integers(300, 6000, (B, 1080, 1920), dtype=np.uint16)
    intr = (910.5, 911.25, 958.7, 541.3)
    for dev_depth in (False, True):
        d = torch.from_numpy(depth.view(np.int16)).cuda() if dev_depth else depth
        bbox, kp3d = engine.pose_post(B, 128, BODY_25.n_parts, TORSO_IDS, d, intr, 1e-3, 2)
        for b in range(B):
            kp = res[b][0]
            m = len(kp)
            assert m == [4, 0, 7][b] or m > 0 or b == 1
            np.testing.assert_array_equal(bbox[b, :m], post_v2.torso_bbox(kp, TORSO_IDS))
            assert (bbox[b, m:] == 0).all() and (kp3d[b, m:] == 0).all()
            for p in range(m):
                want = post_v2.lift_3d(kp[p], depth[b], *intr, 1e-3, 2)
                np.testing.assert_array_equal(kp3d[b, p].view(np.uint64), want.view(np.uint64))
    with pytest.raises(_ffi.PosehipError):
        engine.pose_post(B, 128, BODY_25.n_parts, [99], None)             # torso id out of range
