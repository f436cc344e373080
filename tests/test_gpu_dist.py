"""GPU: the frame-parallel layer (posehip/dist.py) around the REAL engine -- rank-local preprocess + forward + decode,
one gather to rank 0 -- against the same frames decoded directly.  (a) `nccl` (= RCCL) backend with a one-rank
group: the collective code path the 8-GPU run takes; (b) two `gloo` ranks sharing the card: ragged shards,
re-assembly in frame order.  Frames are independent, so the gathered result must equal the direct one bit for bit."""
import os
import socket
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
H, W, N_FRAMES, MAXP = 128, 192, 7, 32


def _frames():
    from posehip import synth
    return synth.synth_frames(4242, N_FRAMES, H, W)


def _make_decode(engine, frames):
    """-> decode_fn(frame_indices) running the hot path of those frames on `engine`."""
    from posehip import _ffi, caffe_io, graph, synth
    from posehip.tables import BODY_25
    net = caffe_io.parse_prototxt(caffe_io.body25_prototxt(H, W))
    plan = graph.build_plan(net, graph.random_weights(net, seed=21))
    engine.build(plan, 4)

    def decode(idx):
        out = []
        for i0 in range(0, len(idx), 4):
            sel = idx[i0:i0 + 4]
            engine.preprocess(frames[sel], _ffi.PH_PRE_NATIVE)
            engine.forward(len(sel))
            real, _ = engine.decode_v2(BODY_25, len(sel), H, W, 1.0, MAXP)
            # A random-weight net may find nobody, and empty records would compare equal whatever the gather does
            # with them: odd frames are decoded from an injected scene of 1 + i % 3 synthetic people instead.
            engine.set_net_out(np.stack([synth.synth_net_output(900 + i, 1 + i % 3, BODY_25, H // 8, W // 8) for i in sel]))
            scene, _ = engine.decode_v2(BODY_25, len(sel), H, W, 1.0, MAXP)
            out += [scene[j] if i % 2 else real[j] for j, i in enumerate(sel)]
        return out
    return decode


def _worker(rank, world, port, backend, q):
    for p in (ROOT, os.path.join(ROOT, "pose-estimation_amd")):
        sys.path.insert(0, p)
    import torch
    import torch.distributed as dist
    from posehip import dist as pdist
    from posehip.engine import Engine
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
        dev = torch.device("cuda", 0)
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
        dev = torch.device("cpu")
    eng = Engine(0)
    res = pdist.run_sharded(_make_decode(eng, _frames()), N_FRAMES, MAXP, 25, dev, block=2)
    if rank == 0:
        q.put([(k.tolist(), s.tolist()) for k, s in res])
    else:
        assert res is None
    dist.barrier()
    eng.close()
    dist.destroy_process_group()


@pytest.mark.timeout(280)
@pytest.mark.parametrize("backend,world", [("nccl", 1), ("gloo", 2)])
def test_run_sharded_on_the_engine_equals_direct_decode(engine, backend, world):
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, backend, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=240)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    want = _make_decode(engine, _frames())(list(range(N_FRAMES)))
    assert len(got) == N_FRAMES
    for (gk, gs), (wk, ws) in zip(got, want):
        np.testing.assert_array_equal(np.asarray(gk, np.float32).reshape(-1, 25, 3), wk[:MAXP])
        np.testing.assert_array_equal(np.asarray(gs, np.float32), ws[:MAXP])
    # the injected scenes (odd frames) must have produced skeletons, in different numbers per frame, so that the
    # packed records, the ragged shards and the re-ordering are compared on real rows
    per_frame = [len(ws) for _, ws in want]
    assert all(per_frame[i] > 0 for i in range(1, N_FRAMES, 2)) and len(set(per_frame[1::2])) > 1, per_frame
