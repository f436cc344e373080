#!/usr/bin/env python3
"""Diagnostic build only: where v1_blobs_kernel spends its cycles (COCO, 1920x1080, B frames of P people).

    make -C pose-estimation_amd BUILD=build_stamp LIB=lib/libposehip_stamp.so EXTRA=-DV1B_STAMP
    POSEHIP_LIB=pose-estimation_amd/lib/libposehip_stamp.so python tools/v1_blobs_stamp.py [P] [B]
"""
import ctypes, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "pose-estimation_amd"))
import numpy as np
from posehip import _ffi, synth
from posehip.engine import Engine
from posehip.tables import COCO_18, MAPIDX, get_body_parts_and_pose_pairs
P = int(sys.argv[1]) if len(sys.argv) > 1 else 60
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
eng = Engine(0)
bp, pp = get_body_parts_and_pose_pairs("COCO")
eng.set_net_out(np.stack([synth.synth_net_output(2000 * P + i, P, COCO_18, 46, 82) for i in range(B)]))
L = _ffi.lib(); cyc = (ctypes.c_ulonglong * 40)()
for _ in range(3):
    eng.decode_v1(bp, pp, MAPIDX, B, 1920, 1080, 0.1, kp_cap=256, person_cap=1024)
if not hasattr(L, "ph_debug_v1_blobs"):      # a build without the stamps: kernel times only (median of 9)
    ts = []
    for _ in range(9):
        eng.decode_v1(bp, pp, MAPIDX, B, 1920, 1080, 0.1, kp_cap=256, person_cap=1024)
        ts.append(eng.decode_v1_times())
    med = [sorted(t[i] for t in ts)[4] for i in range(3)]
    print("kernel ms (runs, blobs, pairs+assemble), median of 9:", [round(t, 3) for t in med], "sum", round(sum(med), 3))
    sys.exit(0)
L.ph_debug_v1_blobs(cyc, 1)
eng.decode_v1(bp, pp, MAPIDX, B, 1920, 1080, 0.1, kp_cap=256, person_cap=1024)
L.ph_debug_v1_blobs(cyc, 0)
print("kernel ms (runs, blobs, pairs+assemble):", [round(t, 3) for t in eng.decode_v1_times()])
nb = max(1, cyc[9])
print(f"blocks {cyc[9]}, runs per block {cyc[10] / nb:.0f}, irregular blobs per block {cyc[5] / nb:.1f}, "
      f"contours walked per block {cyc[6] / nb:.1f}, vertices per contour {cyc[11] / max(1, cyc[6]):.0f}")
print(f"per block: setup {cyc[7] / nb:.0f} cycles, walk phase {cyc[8] / nb:.0f}, whole kernel {cyc[12] / nb:.0f}")
nc = max(1, cyc[6])
print(f"per contour (one group of 8 lanes): trace {cyc[0] / nc:.0f} cycles, fill + outline {cyc[1] / nc:.0f}")
print(f"pixel-by-pixel re-evaluation: {cyc[4]} spans, {cyc[3]} pixels ({cyc[3] / nb:.0f} per block)")
print(f"trace steps per contour {cyc[13] / nc:.0f} ({cyc[0] / max(1, cyc[13]):.0f} cycles per step); fill rows per contour "
      f"{cyc[14] / nc:.0f}; outline {cyc[15] / nc:.0f} cycles per contour")
nf = max(1, cyc[16])
print(f"fast path: {cyc[16]} blobs done, {cyc[17]} without table space, {cyc[18]} given back; per blob: pieces "
      f"{cyc[23] / max(1, cyc[16] + cyc[17] + cyc[18]):.0f}, loops {cyc[25] / nf:.2f}, vertices {cyc[26] / nf:.0f}; cycles: piece list "
      f"{cyc[19] / nf:.0f}, segments {cyc[20] / nf:.0f}, loop walk {cyc[21] / nf:.0f}, placement {cyc[22] / nf:.0f}, "
      f"fills {cyc[24] / nf:.0f}; whole call (all blobs) {cyc[2] / max(1, cyc[5]):.0f}")
print(f"setup, cycles per block from the kernel's start: pieces scattered {cyc[27] / nb:.0f}, rows sorted {cyc[28] / nb:.0f}, "
      f"unions {cyc[29] / nb:.0f}, roots {cyc[30] / nb:.0f}, walks start {cyc[7] / nb:.0f}")
print(f"segments: {cyc[31] / max(1, cyc[16]):.0f} steps per blob over all lanes, {cyc[13] / max(1, cyc[16]):.0f} = the sum over a blob's rounds of the longest segment")
print(f"tables in LDS for {cyc[33]} blobs, a bitmap for {cyc[32]}")
print(f"segment rounds, cycles per blob: longest node set-up {cyc[34] / nf:.0f}, longest step loop {cyc[35] / nf:.0f}, whole rounds {cyc[36] / nf:.0f}")
print(f"node set-up per round: bitmap blobs {cyc[37] / max(1, cyc[3]):.0f} cycles ({cyc[3]} rounds), LDS tables without bitmap {cyc[38] / max(1, cyc[4]):.0f} ({cyc[4]}), tables in the pool {cyc[39] / max(1, cyc[6]):.0f} ({cyc[6]})")
