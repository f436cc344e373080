"""GPU: the allocation-free decode entry point (ph_decode_v2_noalloc, include/posehip.h) against the existing
ph_decode_v2, through the C ABI, bit-exact -- two calls in a row, the second with a smaller input (fewer frames, a
smaller map), so stale workspace contents of the first call would show."""
import numpy as np
import pytest

from posehip import _ffi, synth
from posehip.tables import BODY_25

pytestmark = pytest.mark.gpu


def _noalloc(engine, frames, Ht, Wt, scale, max_people):
    import torch
    B = frames.shape[0]
    dev = torch.device("cuda:0")
    net = torch.from_numpy(frames).to(dev)
    # outputs poisoned so that rows the kernels do not write cannot pass for results
    n = torch.full((B,), -7, dtype=torch.int32, device=dev)
    kp = torch.full((B, max_people, BODY_25.n_parts, 3), float("nan"), dtype=torch.float32, device=dev)
    sc = torch.full((B, max_people), float("nan"), dtype=torch.float32, device=dev)
    ovf = torch.full((1,), -7, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    engine.decode_v2_noalloc(BODY_25, net, Ht, Wt, n, kp, sc, scale=scale, overflow=ovf)
    engine.sync()
    return n.cpu().numpy(), kp.cpu().numpy(), sc.cpu().numpy(), int(ovf.cpu()[0])


def _reference(engine, frames, Ht, Wt, scale, max_people):
    engine.set_net_out(frames)
    res, extras = engine.decode_v2(BODY_25, frames.shape[0], Ht, Wt, scale=scale, max_people=max_people)
    return res, extras["n_people"]


def _compare(engine, frames, Ht, Wt, scale=1.5, max_people=128):
    n, kp, sc, ovf = _noalloc(engine, frames, Ht, Wt, scale, max_people)
    res, n_ref = _reference(engine, frames, Ht, Wt, scale, max_people)
    assert ovf == 0
    np.testing.assert_array_equal(n, n_ref)
    for b, (kp_r, sc_r) in enumerate(res):
        m = len(sc_r)
        np.testing.assert_array_equal(kp[b, :m], kp_r)
        np.testing.assert_array_equal(sc[b, :m], sc_r)
    return int(n.sum())


def test_noalloc_decode_equals_decode_twice_in_a_row(engine):
    big = np.stack([synth.synth_net_output(300 + i, p, BODY_25, 46, 82) for i, p in enumerate([6, 20, 0, 3])])
    assert _compare(engine, big, 368, 656) > 0
    small = np.stack([synth.synth_net_output(400, 2, BODY_25, 46, 46)])       # fewer frames, smaller maps
    assert _compare(engine, small, 368, 368) > 0
    # and the oracle agrees with what both entry points returned for the small call
    from oracle import decode_v2
    kp_o, sc_o = decode_v2.decode(small[0], BODY_25, 368, 368, 1.5)
    n, kp, sc, _ = _noalloc(engine, small, 368, 368, 1.5, 128)
    np.testing.assert_array_equal(kp[0, :len(sc_o)], kp_o)
    np.testing.assert_array_equal(sc[0, :len(sc_o)], sc_o)


def test_noalloc_rejects_what_does_not_fit(engine):
    import torch
    B = _ffi.PH_NOALLOC_MAX_FRAMES + 1
    dev = torch.device("cuda:0")
    net = torch.zeros((B, 78, 46, 82), dtype=torch.float32, device=dev)
    n = torch.zeros((B,), dtype=torch.int32, device=dev)
    kp = torch.zeros((B, 128, 25, 3), dtype=torch.float32, device=dev)
    sc = torch.zeros((B, 128), dtype=torch.float32, device=dev)
    with pytest.raises(_ffi.PosehipError, match=r"\(-3\).*static scratch"):
        engine.decode_v2_noalloc(BODY_25, net, 368, 656, n, kp, sc)
    # the largest call that fits does run
    frames = np.stack([synth.synth_net_output(500 + i, 1 + i % 3, BODY_25, 46, 82)
                       for i in range(_ffi.PH_NOALLOC_MAX_FRAMES)])
    assert _compare(engine, frames, 368, 656) > 0
