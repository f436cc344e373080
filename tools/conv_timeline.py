#!/usr/bin/env python3
"""Diagnostic build only: where the time of ONE short-K 3x3 launch goes, per block (s_memrealtime, 100 MHz).

    cd pose-estimation_amd && hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -DPH_TIMELINE \
        -c csrc/conv_mfma.hip -o build_diag/conv_mfma.o && hipcc -shared -fPIC --offload-arch=gfx950 \
        -o lib/libposehip_tl.so build/capi.o build_diag/conv_mfma.o build/preprocess.o build/decode_v2.o \
        build/decode_v1.o build/post_v2.o build/movenet.o
    POSEHIP_LIB=pose-estimation_amd/lib/libposehip_tl.so python tools/conv_timeline.py [cin cout]
"""
import ctypes, os, sys
sys.path.insert(0, "pose-estimation_amd")
import numpy as np
from posehip import _ffi, caffe_io, graph, synth
from posehip.engine import Engine
cin = int(sys.argv[1]) if len(sys.argv) > 1 else 128
cout = int(sys.argv[2]) if len(sys.argv) > 2 else 128
eng = Engine(0); B = 32
H, W, k = 46, 82, 3
txt = f'input: "image"\ninput_dim: 1\ninput_dim: 3\ninput_dim: {H}\ninput_dim: {W}\n'
txt += caffe_io._conv("c0", "image", cin, 1) + caffe_io._act("ReLU", "r0", "c0")
txt += caffe_io._conv("c1", "c0", cout, k) + caffe_io._act("ReLU", "r1", "c1")
txt += caffe_io._conv("out", "c1", 8, 1)
net = caffe_io.parse_prototxt(txt)
plan = graph.build_plan(net, graph.random_weights(net, 0, out_gain=1.0))
eng.build(plan, B)
eng.preprocess(synth.synth_frames(0, B, H, W), _ffi.PH_PRE_CV2)
L = ctypes.CDLL(os.environ["POSEHIP_LIB"])
import time
t0 = time.time()
while time.time() - t0 < 2.0: eng.forward_timed(B)
ms = eng.forward_timed(B); eng.sync()
nblk = 504 * ((cout + 127) // 128)
buf = (ctypes.c_ulonglong * (nblk * 5))()
assert L.ph_debug_conv_timeline(buf, nblk) == 0
t = np.array(buf, dtype=np.int64).reshape(nblk, 5)
hw = t[:, 4]; t = t[:, :4].astype(np.float64)
base = t[:, 0].min()
t = (t - base) / 100.0   # us
print(f"kernel (HIP events) {ms[1] * 1e3:.1f} us; blocks {nblk}")
for name, col in (("entry", 0), ("K loop start", 1), ("K loop end", 2), ("last store issued", 3)):
    v = t[:, col]
    print(f"{name:18s} min {v.min():6.2f}  p10 {np.percentile(v, 10):6.2f}  median {np.median(v):6.2f}  p90 {np.percentile(v, 90):6.2f}  max {v.max():6.2f} us")
d = np.diff(t, axis=1)
for name, col in (("prologue", 0), ("K loop", 1), ("epilogue", 2)):
    v = d[:, col]
    print(f"{name:18s} min {v.min():6.2f}  median {np.median(v):6.2f}  max {v.max():6.2f} us")
cu = ((hw >> 8) & 0xf) | (((hw >> 12) & 0x3) << 4) | (((hw >> 13) & 0x7) << 6)   # cu_id, sh_id, se_id (approximate)
print("distinct (se, sh, cu) ids seen:", len(np.unique(cu)), " blocks per id: max", np.bincount(np.unique(cu, return_inverse=True)[1]).max())
kl = d[:, 1]
bid = np.arange(nblk)
print("K loop us by XCD (blockIdx & 7): ", [f"{np.median(kl[bid % 8 == x]):.1f}/{kl[bid % 8 == x].max():.1f}" for x in range(8)])
print("K loop us histogram:", np.histogram(kl, bins=8)[0].tolist(), [round(float(x), 1) for x in np.histogram(kl, bins=8)[1]])
slow = kl > (kl.min() + kl.max()) / 2
print("slow blocks:", int(slow.sum()), " by XCD:", [int(slow[bid % 8 == x].sum()) for x in range(8)])
ids, inv = np.unique(cu, return_inverse=True)
print("slow blocks by (se,sh,cu) id:", [int(slow[inv == i].sum()) for i in range(len(ids))])
print("entry us of slow / fast blocks (median):", round(float(np.median(t[slow, 0])), 2) if slow.any() else None, round(float(np.median(t[~slow, 0])), 2))
print("first 64 blocks K loop us:", [round(float(x), 1) for x in kl[:64]])
# pairs sharing a CU: same XCD (bid & 7) and same id
key = (bid % 8) * 4096 + cu
u, inv2, cnt = np.unique(key, return_inverse=True, return_counts=True)
print("blocks per (XCD, CU) histogram:", np.bincount(cnt).tolist())
alone = cnt[inv2] == 1
print("K loop median us: alone on a CU", round(float(np.median(kl[alone])), 1) if alone.any() else None, " sharing", round(float(np.median(kl[~alone])), 1))
# per-wave epilogue stamps (first min(nblk, 1024) blocks)
nb = min(nblk, 1024)
bw = (ctypes.c_ulonglong * (nb * 64))()
if hasattr(L, "ph_debug_conv_timeline_waves") and L.ph_debug_conv_timeline_waves(bw, nb) == 0:
    w = np.array(bw, dtype=np.int64).reshape(nb, 8, 8).astype(np.float64)
    t0 = w[:, :, 7].min(axis=1)[:, None, None]
    w = (w - t0) / 100.0
    names = ["last step done", "re-fetches landed", "bias+act done", "barrier passed", "row table done", "staged", "stores issued"]
    print("per-wave epilogue stamps, us from block entry (median over blocks; per wave 0..7):")
    for k, nme in enumerate(names):
        print(f"  {nme:18s}", [round(float(np.median(w[:, wv, k])), 2) for wv in range(8)])
    print("  barrier wait per wave (us, median):", [round(float(np.median(w[:, wv, 3] - w[:, wv, 2])), 2) for wv in range(8)])
    sl = kl[:nb] > (kl.min() + kl.max()) / 2
    for nme, sel in (("fast blocks", ~sl), ("slow blocks", sl)):
        if sel.any():
            print(f"  {nme}: last step done -> stores issued (us, median over waves & blocks)",
                  [round(float(np.median(w[sel][:, :, k])), 2) for k in range(7)])
