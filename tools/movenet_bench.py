#!/usr/bin/env python3
"""BASELINE configs[3]: MoveNet MultiPose-Lightning-shaped head decode (256x256 input -> 64x64 grid), batch 64,
plus the in-tree crop_and_resize / crop-region update steps, on device-resident synthetic tensors.

    python tools/movenet_bench.py [--batch 64] [--reps 20]
Prints device ms (HIP events on the engine's stream via ph_sync-bracketed host timing of on_device calls) and the
algorithmic HBM rate: heads read once = 88 floats per grid cell; crop = 12 bytes written + <= 12 bytes read per
output pixel."""
import argparse
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pose-estimation_amd"))

from posehip import _ffi, movenet  # noqa: E402


def timed(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--reps", type=int, default=20)
    args = ap.parse_args()
    B, G, P = args.batch, 64, 6
    e = movenet._eng()
    L, ctx = e._lib, e._ctx
    g = torch.Generator(device="cuda").manual_seed(0)
    center = torch.rand((B, G, G), device="cuda", generator=g) ** 4
    heat = torch.rand((B, G, G, 17), device="cuda", generator=g)
    regr = torch.randn((B, G, G, 34), device="cuda", generator=g) * 8
    offs = torch.rand((B, G, G, 34), device="cuda", generator=g) - 0.5
    box = torch.rand((B, G, G, 2), device="cuda", generator=g) * 30 + 1
    out = torch.empty((B, P, 56), device="cuda")
    p = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731

    def decode():
        _ffi.check(L.ph_movenet_multipose_decode(ctx, p(center), p(heat), p(regr), p(offs), p(box), B, G, P,
                                                 C.c_float(0.3), p(out), 1), "decode")

    ms = timed(decode, args.reps)
    head_bytes = B * G * G * 88 * 4
    print(f"multipose decode  B={B} {G}x{G} grid, {P} people: {ms:.3f} ms = {B / ms * 1e3:.0f} frames/s, "
          f"{head_bytes / ms / 1e6:.0f} GB/s of head bytes ({head_bytes / 1e6:.0f} MB, read once) "
          f"[{100 * head_bytes / ms / 1e6 / 8000:.1f} % of 8 TB/s]")

    H, W, S = 1080, 1920, 256
    frames = torch.randint(0, 256, (B, H, W, 3), device="cuda", dtype=torch.uint8, generator=g)
    crop = torch.tensor(np.stack([movenet.region_to_array(movenet.init_crop_region(H, W))] * B), device="cuda")
    crop_out = torch.empty_like(crop)
    inp = torch.empty((B, S, S, 3), device="cuda", dtype=torch.int32)
    kp = torch.rand((B, 17, 3), device="cuda", generator=g)

    def crop_fn():
        _ffi.check(L.ph_movenet_crop_resize(ctx, p(frames), B, H, W, p(crop), S, p(inp), 1), "crop")

    def update_fn():
        _ffi.check(L.ph_movenet_update(ctx, p(kp), p(crop), B, H, W, p(crop_out), 1), "update")

    ms = timed(crop_fn, args.reps)
    touched = B * S * S * (12 + 12)
    print(f"crop_and_resize   B={B} {W}x{H} -> {S}x{S}: {ms:.3f} ms, {touched / ms / 1e6:.0f} GB/s of written + gathered bytes")
    ms = timed(update_fn, args.reps)
    print(f"update + next crop B={B}: {ms:.3f} ms (one thread per stream, latency-bound)")


if __name__ == "__main__":
    main()
