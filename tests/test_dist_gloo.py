"""CPU: the N>1 path (frame sharding + gather to rank 0) with world_size 2 on gloo."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

from posehip import dist as pdist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_frames_partition():
    for n, world, block in [(0, 2, 1), (7, 2, 1), (64, 8, 32), (65, 4, 8), (5, 8, 1)]:
        seen = []
        for r in range(world):
            mine = pdist.shard_frames(n, r, world, block)
            assert mine == sorted(mine)
            seen += mine
        assert sorted(seen) == list(range(n))
    assert pdist.shard_frames(8, 1, 2, 2) == [2, 3, 6, 7]
    with pytest.raises(ValueError):
        pdist.shard_frames(4, 2, 2)


def test_pad_results():
    res = [(np.ones((2, 25, 3), np.float32), np.float32([.5, .4])), (None, None),
           (np.ones((5, 25, 3), np.float32), np.ones(5, np.float32))]
    cnt, kp, sc = pdist.pad_results(res, 4, 25)
    assert list(cnt) == [2, 0, 4] and kp.shape == (3, 4, 25, 3) and sc[2].sum() == 4


def _fake_decode(frame_ids):
    out = []
    for f in frame_ids:
        m = f % 3
        kp = np.full((m, 25, 3), float(f), np.float32)
        out.append((kp, np.full((m,), f / 10.0, np.float32)))
    return out


def _worker(rank, world, port, n_frames, q):
    sys.path.insert(0, os.path.join(ROOT, "pose-estimation_amd"))
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    res = pdist.run_sharded(_fake_decode, n_frames, 4, 25, torch.device("cpu"), block=2)
    if rank == 0:
        q.put([(k.tolist(), s.tolist()) for k, s in res])
    else:
        assert res is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_world2_gather_reassembles_in_frame_order():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    n_frames = 7                                  # ragged: ranks get 4 and 3 frames
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_frames, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=90)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = _fake_decode(list(range(n_frames)))    # same per-frame output regardless of world size
    assert len(got) == n_frames
    for (gk, gs), (wk, ws) in zip(got, want):
        assert np.array_equal(np.asarray(gk, np.float32).reshape(-1, 25, 3), wk)
        assert np.array_equal(np.asarray(gs, np.float32), ws)
