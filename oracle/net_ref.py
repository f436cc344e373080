"""ORACLE (test infrastructure) -- network forward on the CPU.

Plain PyTorch fp32 execution of the Caffe deploy graph the reference runs with
``cv2.dnn`` (``openpose/opencv/skeleton.py:66-75``: blobFromImage -> setInput ->
forward) or Caffe inside OpenPose (``openpose/native/python/skeleton.py:447``).
Layer semantics restated: Convolution (stride 1, zero pad), ReLU, PReLU
(per-channel or shared slope), Pooling MAX 2x2/2 (ceil mode), Concat axis 1.

PARITY UNPINNED vs the real engines: neither cv2 nor Caffe nor any model file
exists offline; this is the fp32 reference the bf16 MFMA path is compared with
(tolerance in tests/test_gpu_net.py).  `bf16_sim=True` additionally rounds
weights and every stored activation to bf16, reproducing the GPU path's storage
precision so layer-level comparisons can be tight.
"""
import numpy as np
import torch
import torch.nn.functional as F


def _bf16(t):
    return t.to(torch.bfloat16).to(torch.float32)


def _one(d, key, default=None):
    v = d.get(key)
    return v[0] if v else default


def forward(net, weights, x, bf16_sim=False, return_blobs=False):
    """net: parsed prototxt (posehip.caffe_io.parse_prototxt); weights:
    {layer: [blobs]}; x: float32 [B,3,H,W] already normalised.  -> f32 NCHW."""
    if bf16_sim:
        # the GPU path stores every fused conv(+activation) result as bf16,
        # except the network outputs which stay fp32
        return _forward_bf16(net, weights, x, return_blobs)
    blobs = {}
    in_name = _one(net, "input")
    layers = net.get("layer", net.get("layers", []))
    for l in layers:
        if _one(l, "type") == "Input":
            in_name = l["top"][0]
    blobs[in_name] = torch.as_tensor(np.asarray(x, np.float32))
    last = in_name
    conv_tops = set()
    with torch.no_grad():
        for l in layers:
            typ, name = _one(l, "type"), _one(l, "name")
            if typ == "Input":
                continue
            bottoms, top = l.get("bottom", []), l["top"][0]
            if typ == "Convolution":
                p = l["convolution_param"][0]
                k = _one(p, "kernel_size")
                pad = _one(p, "pad", 0)
                w = torch.as_tensor(np.asarray(weights[name][0], np.float32))
                w = w.reshape(_one(p, "num_output"), -1, k, k)
                b = torch.as_tensor(np.asarray(weights[name][1], np.float32)).reshape(-1) \
                    if len(weights[name]) > 1 and _one(p, "bias_term", True) else None
                y = F.conv2d(blobs[bottoms[0]], w, b, stride=1, padding=pad)
                conv_tops.add(top)
            elif typ == "ReLU":
                y = F.relu(blobs[bottoms[0]])
            elif typ == "PReLU":
                s = torch.as_tensor(np.asarray(weights[name][0], np.float32)).reshape(-1)
                y = F.prelu(blobs[bottoms[0]], s)
            elif typ == "Pooling":
                y = F.max_pool2d(blobs[bottoms[0]], 2, 2, ceil_mode=True)
            elif typ == "Concat":
                y = torch.cat([blobs[b] for b in bottoms], dim=1)
            else:
                raise ValueError(f"unsupported layer type {typ}")
            blobs[top] = y
            last = top
    out = blobs[last].numpy()
    if return_blobs:
        return out, {k: v.numpy() for k, v in blobs.items()}
    return out


def _forward_bf16(net, weights, x, return_blobs):
    layers = net.get("layer", net.get("layers", []))
    in_name = _one(net, "input")
    for l in layers:
        if _one(l, "type") == "Input":
            in_name = l["top"][0]
    # which conv tops are followed by an in-place activation
    has_act = set()
    for l in layers:
        if _one(l, "type") in ("ReLU", "PReLU"):
            has_act.add(l["bottom"][0])
    # network-output members keep fp32
    last = layers[-1]
    finals = set(last.get("bottom", [])) if _one(last, "type") == "Concat" else {last["top"][0]}
    blobs = {in_name: _bf16(torch.as_tensor(np.asarray(x, np.float32)))}
    last_top = in_name
    with torch.no_grad():
        for l in layers:
            typ, name = _one(l, "type"), _one(l, "name")
            if typ == "Input":
                continue
            bottoms, top = l.get("bottom", []), l["top"][0]
            if typ == "Convolution":
                p = l["convolution_param"][0]
                k = _one(p, "kernel_size")
                w = _bf16(torch.as_tensor(np.asarray(weights[name][0], np.float32)).reshape(
                    _one(p, "num_output"), -1, k, k))
                b = torch.as_tensor(np.asarray(weights[name][1], np.float32)).reshape(-1) \
                    if len(weights[name]) > 1 and _one(p, "bias_term", True) else None
                y = F.conv2d(blobs[bottoms[0]], w, b, stride=1, padding=_one(p, "pad", 0))
                if top not in has_act and top not in finals:
                    y = _bf16(y)
            elif typ in ("ReLU", "PReLU"):
                y = blobs[bottoms[0]]
                if typ == "ReLU":
                    y = F.relu(y)
                else:
                    y = F.prelu(y, torch.as_tensor(np.asarray(weights[name][0], np.float32)).reshape(-1))
                if top not in finals:
                    y = _bf16(y)
            elif typ == "Pooling":
                y = F.max_pool2d(blobs[bottoms[0]], 2, 2, ceil_mode=True)
            elif typ == "Concat":
                y = torch.cat([blobs[b] for b in bottoms], dim=1)
            else:
                raise ValueError(f"unsupported layer type {typ}")
            blobs[top] = y
            last_top = top
    out = blobs[last_top].numpy()
    if return_blobs:
        return out, {k: v.numpy() for k, v in blobs.items()}
    return out
