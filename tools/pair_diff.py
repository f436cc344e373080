"""Debug aid: per-stage difference between the pair kernel and the separate 1x1 launches."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "pose-estimation_amd"))
from posehip import _ffi, caffe_io, graph, synth
from posehip.engine import Engine

H, W, B = 64, 96, 3
net = caffe_io.parse_prototxt(caffe_io.body25_prototxt(H, W))
weights = graph.random_weights(net, seed=17)
plan = graph.build_plan(net, weights)
eng = Engine(0)
eng.build(plan, B)
frames = synth.synth_frames(9, B, H, W)
res = []
for flag in ("0", "1"):
    os.environ["POSEHIP_PAIR_KERNEL"] = flag
    names = eng.op_kernels()
    eng.preprocess(frames, _ffi.PH_PRE_NATIVE)
    eng.forward(B)
    out = eng.net_out(B)
    bufs = {}
    for i, op in enumerate(plan.ops):
        if op.get("k", 0) == 1 and op.get("cout", 999) <= 64:
            for (b, c) in op["dst"]:
                bufs[(i, plan.op_names[i], b, c)] = eng.get_buffer(b).astype(np.float32)
    res.append((out, bufs))
print("net_out maxdiff", np.abs(res[0][0] - res[1][0]).max(), "scale", np.abs(res[0][0]).max())
for k in res[0][1]:
    a, b = res[0][1][k], res[1][1][k]
    c0, n = k[3], plan.ops[k[0]]["cout"]
    d = np.abs(a - b)
    print(k, a.shape, "diff in own window", d[..., c0:c0 + n].max() if a.ndim == 4 else d.max(), "scale", np.abs(a).max(),
          "n differing", int((d > 0).sum()), "of", d.size)
