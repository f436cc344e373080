#!/usr/bin/env python3
"""Achievable HBM bandwidth of a plain streaming copy on this GPU (read + write bytes / time)."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pose-estimation_amd"))
from posehip import _ffi  # noqa: E402

lib = _ffi.lib()
lib.ph_debug_copy_bandwidth.argtypes = [C.c_size_t, C.c_int, C.POINTER(C.c_float)]
for gib in (1, 4):
    out = C.c_float()
    rc = lib.ph_debug_copy_bandwidth(gib << 30, 10, C.byref(out))
    print(f"copy {gib} GiB x10: rc={rc}  {out.value / 1e3:.2f} TB/s (read+write)")
