"""Several batches in flight on one GPU.

Frames of different batches never interact, so batch k+1's convolutions can run while batch k is
still in its latency-bound decode kernels and in the partial last wave of its conv launches.  `BatchPipeline`
keeps `depth` engines (one ``ph_ctx`` = one HIP stream + its own activation buffers each) over one plan and
deals submitted batches to them round-robin; results come back in submission order.  This is the batched
counterpart of calling ``PyOpenPoseNative.predict`` in a loop (openpose/native/python/skeleton.py:444-450) and
what ``bench.py`` times (there hand-rolled, because it also orders an RCCL gather behind each batch).

Frames that arrive in host memory (a video stream: the "data preparation" stage of the reference's pipeline,
openpose/native/python/inference.py:197-247, hands NumPy arrays to the estimator) go through `submit_host`: a
dedicated copy stream and two device frame buffers per engine, so the H2D copy of batch k+1 runs while batch k
computes and is handed over by an event -- nothing on the host waits for a copy.  Call `ensure_hw_queues(depth + 2)`
before the process touches the GPU, so that the copy stream gets a hardware queue of its own.
"""
import numpy as np

from . import _ffi
from .engine import Engine


def ensure_hw_queues(n_streams, limit=8):
    """HIP spreads a process's streams over GPU_MAX_HW_QUEUES hardware queues (default 4).  Streams that share a
    queue run their commands one after the other -- a 3.5 ms frame copy then stalls the engine that shares its
    queue (bench.py: 0.82 of the resident rate with 4 queues, 0.997 with 8).  Raises the variable to `n_streams`
    (at most `limit`) if it is lower; must run before the first HIP call of the process (the runtime reads it once).
    Returns the value in effect for a process that has not initialised HIP yet."""
    import os
    cur = int(os.environ.get("GPU_MAX_HW_QUEUES", "") or 4)
    want = min(max(cur, int(n_streams)), max(cur, int(limit)))
    if want != cur:
        os.environ["GPU_MAX_HW_QUEUES"] = str(want)
    return want


class BatchPipeline:
    def __init__(self, plan, batch, model, net_h, net_w, device=0, depth=2, max_people=128,
                 pre_mode=_ffi.PH_PRE_NATIVE, scale=1.0):
        import torch
        if not torch.cuda.is_available():
            raise _ffi.PosehipError("BatchPipeline needs a GPU (no CPU fallback)")
        self._torch = torch
        self.batch, self.model, self.net_h, self.net_w = batch, model, net_h, net_w
        self.max_people, self.pre_mode, self.scale = max_people, pre_mode, scale
        dev = torch.device("cuda", device)
        self.streams = [torch.cuda.Stream(device=dev) for _ in range(max(1, depth))]
        self.engines = []
        self.out = []
        J = model.n_parts
        for ts in self.streams:
            e = Engine(device, ts.cuda_stream)
            e.build(plan, batch)
            self.engines.append(e)
            self.out.append((torch.empty((batch,), dtype=torch.int32).pin_memory(),
                             torch.empty((batch, max_people, J, 3), dtype=torch.float32).pin_memory(),
                             torch.empty((batch, max_people), dtype=torch.float32).pin_memory()))
        self._pending = [None] * len(self.streams)   # per engine: (ticket, B) of the batch it is running
        self._copy_stream = torch.cuda.Stream(device=dev)
        self._dev = dev
        # per engine: two device frame buffers [buffer, copied event, event of the preprocess that last read it]
        self._stage = [[[None, None, None], [None, None, None]] for _ in self.streams]
        self._flip = [0] * len(self.streams)
        self._done = {}
        self._next = 0

    def _collect(self, k):
        if self._pending[k] is None:
            return
        self.streams[k].synchronize()
        ticket, B = self._pending[k]
        self._pending[k] = None
        self.engines[k].decode_v2_status()    # raises (-2) if this batch's decode dropped people
        n, kp, sc = self.out[k]
        self._done[ticket] = (n[:B].numpy().copy(), kp[:B].numpy().copy(), sc[:B].numpy().copy())

    def submit_host(self, frames):
        """frames: u8 [B<=batch, H0, W0, 3] BGR in HOST memory -- a pinned torch tensor (copied asynchronously; it must
        not be rewritten before `result` of this ticket or of a later one on the same engine) or a NumPy array
        (staged through a pinned copy first).  Returns a ticket; copy and compute are only enqueued."""
        torch = self._torch
        if isinstance(frames, np.ndarray):
            frames = torch.from_numpy(np.ascontiguousarray(frames, dtype=np.uint8)).pin_memory()
        if frames.is_cuda or frames.dtype != torch.uint8 or frames.dim() != 4 or frames.shape[3] != 3:
            raise ValueError("submit_host takes host u8 frames [B,H,W,3]")
        B = int(frames.shape[0])
        if B < 1 or B > self.batch:
            raise ValueError(f"batch {B} outside [1, {self.batch}]")
        k = self._next % len(self.engines)
        j = self._flip[k]
        self._flip[k] ^= 1
        slot = self._stage[k][j]
        if slot[0] is None or tuple(slot[0].shape[1:]) != tuple(frames.shape[1:]):
            slot[0] = torch.empty((self.batch,) + tuple(frames.shape[1:]), dtype=torch.uint8, device=self._dev)
            slot[1] = torch.cuda.Event()
        with torch.cuda.stream(self._copy_stream):
            if slot[2] is not None:
                self._copy_stream.wait_event(slot[2])     # the preprocess that read this buffer two batches ago
            slot[0][:B].copy_(frames.contiguous(), non_blocking=True)
            slot[1].record(self._copy_stream)
        return self._submit(slot[0][:B], k, copied=slot[1], slot=slot)

    def submit(self, frames):
        """frames: contiguous CUDA u8 tensor [B<=batch, H0, W0, 3] BGR (must stay alive until `result`).
        Returns a ticket; the work is only enqueued."""
        B = int(frames.shape[0])
        if B < 1 or B > self.batch:
            raise ValueError(f"batch {B} outside [1, {self.batch}]")
        return self._submit(frames, self._next % len(self.engines))

    def _submit(self, frames, k, copied=None, slot=None):
        B = int(frames.shape[0])
        self._collect(k)                      # the batch this engine ran `depth` submissions ago
        e = self.engines[k]
        if copied is not None:
            self.streams[k].wait_event(copied)
        else:
            self.streams[k].wait_stream(self._torch.cuda.current_stream())   # frames were produced on the caller's stream
        e.preprocess(frames, self.pre_mode)
        if slot is not None:
            if slot[2] is None:
                slot[2] = self._torch.cuda.Event()
            slot[2].record(self.streams[k])
        e.forward(B)
        e.decode_v2_dev(self.model, B, self.net_h, self.net_w, self.scale, self.max_people)
        e.decode_v2_results(*self.out[k])
        ticket = self._next
        self._pending[k] = (ticket, B)
        self._next += 1
        return ticket

    def result(self, ticket):
        """(n_people i32[B], keypoints f32[B, max_people, J, 3], scores f32[B, max_people]) of a ticket."""
        if ticket not in self._done:
            for k, p in enumerate(self._pending):
                if p is not None and p[0] == ticket:
                    self._collect(k)
        if ticket not in self._done:
            raise KeyError(f"unknown or already collected ticket {ticket}")
        return self._done.pop(ticket)

    def close(self):
        for k in range(len(self.engines)):
            self._collect(k)
        for e in self.engines:
            e.close()
        self.engines = []
