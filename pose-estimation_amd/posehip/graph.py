"""Caffe layer graph -> fused MI355X execution plan.

Input: a parsed deploy prototxt (``caffe_io.parse_prototxt``) and a
``{layer: [blobs]}`` weight dict -- the two files the reference hands to
``cv2.dnn.readNetFromCaffe`` (``openpose/opencv/skeleton.py:50``).  Output: the
``ph_buf_desc`` / ``ph_op_desc`` arrays of include/posehip.h:

* Convolution + in-place ReLU/PReLU -> one fused op (bias + activation in the
  MFMA epilogue);
* Concat -> nothing: members are written straight into 8-aligned channel windows
  of the consumer's buffer (a producer feeding several concats gets several
  destinations), padding channels stay zero and get zero weights through
  ``cin_map``;
* the network output concat -> fp32 NCHW windows of the net-output tensor;
* activations are bf16 NHWC with a zero border as wide as the largest consumer
  kernel's padding, so the kernels never mask.
"""
import ctypes as C
import os

import numpy as np

from . import _ffi


def _r8(c):
    return (c + 7) // 8 * 8


class _Node:
    def __init__(self, kind, name, cout, h, w):
        self.kind = kind          # input | conv | pool | concat
        self.name = name
        self.cout = cout
        self.h, self.w = h, w
        self.src = []             # input nodes
        self.k = 1
        self.act = _ffi.PH_ACT_NONE
        self.act_name = None
        self.bias_term = True
        self.consumers = []
        # placement
        self.views = []           # [(buf, coff)] bf16 locations of this blob
        self.f32_coff = -1


class Plan:
    """Execution plan + the numpy arrays its descriptors point into."""

    def __init__(self):
        self.bufs = []            # dicts: channels,height,width,pad
        self.ops = []             # dicts (see _op_struct)
        self.out_channels = self.out_h = self.out_w = 0
        self.in_h = self.in_w = 0
        self.op_names = []
        self._keep = []

    def to_ctypes(self):
        bufs = (_ffi.PhBufDesc * len(self.bufs))()
        for i, b in enumerate(self.bufs):
            bufs[i] = _ffi.PhBufDesc(b["channels"], b["height"], b["width"], b["pad"])
        ops = (_ffi.PhOpDesc * len(self.ops))()
        keep = []
        for i, o in enumerate(self.ops):
            d = ops[i]
            d.type = o["type"]
            d.in_buf, d.in_coff, d.cin_read = o["in_buf"], o["in_coff"], o["cin_read"]
            d.kh = d.kw = o["k"]
            d.cout, d.act = o["cout"], o["act"]
            d.n_dst = len(o["dst"])
            for k, (b, c) in enumerate(o["dst"]):
                d.dst_buf[k], d.dst_coff[k] = b, c
            d.f32_coff = o["f32_coff"]
            d.tile_n = o.get("tile_n", 0)
            d.pool = o.get("pool", 0)
            if o["type"] == _ffi.PH_OP_CONV:
                w = np.ascontiguousarray(o["weight"], dtype=np.float32)
                d.cin_caffe = w.shape[1]
                d.weight = w.ctypes.data_as(C.POINTER(C.c_float))
                keep.append(w)
                if o["bias"] is not None:
                    b = np.ascontiguousarray(o["bias"], dtype=np.float32)
                    d.bias = b.ctypes.data_as(C.POINTER(C.c_float))
                    keep.append(b)
                if o["slope"] is not None:
                    s = np.ascontiguousarray(o["slope"], dtype=np.float32).reshape(-1)
                    d.slope = s.ctypes.data_as(C.POINTER(C.c_float))
                    d.slope_count = s.size
                    keep.append(s)
                m = np.ascontiguousarray(o["cin_map"], dtype=np.int32)
                d.cin_map = m.ctypes.data_as(C.POINTER(C.c_int32))
                keep.append(m)
        self._keep = keep
        return bufs, ops

    def flops_per_frame(self):
        """Algorithmic conv FLOPs (2*k*k*Cin*Cout*H*W, unpadded channels)."""
        tot = 0.0
        for o in self.ops:
            if o["type"] == _ffi.PH_OP_CONV:
                tot += 2.0 * o["k"] ** 2 * o["weight"].shape[1] * o["cout"] * o["h"] * o["w"]
        return tot


def _layers(net):
    return net.get("layer", net.get("layers", []))


def _one(d, key, default=None):
    v = d.get(key)
    return v[0] if v else default


def input_shape(net):
    """(C, H, W) declared by the prototxt."""
    if "input_dim" in net:
        d = net["input_dim"]
        return d[1], d[2], d[3]
    if "input_shape" in net:
        d = net["input_shape"][0]["dim"]
        return d[1], d[2], d[3]
    for l in _layers(net):
        if _one(l, "type") == "Input":
            d = l["input_param"][0]["shape"][0]["dim"]
            return d[1], d[2], d[3]
    raise ValueError("prototxt declares no input shape")


def _build_nodes(net, height, width):
    c_in, h0, w0 = input_shape(net)
    height = height or h0
    width = width or w0
    blobs = {}
    nodes = []
    in_name = _one(net, "input")
    for l in _layers(net):
        if _one(l, "type") == "Input":
            in_name = l["top"][0]
    if in_name is None:
        raise ValueError("prototxt has no input blob")
    inp = _Node("input", in_name, c_in, height, width)
    blobs[in_name] = inp
    nodes.append(inp)
    last = inp
    for l in _layers(net):
        typ = _one(l, "type")
        name = _one(l, "name")
        bottoms = l.get("bottom", [])
        tops = l.get("top", [])
        if typ == "Input":
            continue
        for b in bottoms:
            if b not in blobs:
                raise ValueError(f"layer {name!r}: unknown bottom {b!r}")
        if typ in ("Convolution", 4):
            p = l["convolution_param"][0]
            k = _one(p, "kernel_size", _one(p, "kernel_h"))
            pad = _one(p, "pad", _one(p, "pad_h", 0))
            stride = _one(p, "stride", 1)
            if _one(p, "kernel_w", k) != k or stride != 1 or k % 2 == 0 or pad != (k - 1) // 2 \
                    or _one(p, "group", 1) != 1 or _one(p, "dilation", 1) != 1:
                raise ValueError(f"layer {name!r}: only odd square stride-1 'same' convolutions are supported")
            src = blobs[bottoms[0]]
            n = _Node("conv", name, _one(p, "num_output"), src.h, src.w)
            n.k = k
            n.bias_term = _one(p, "bias_term", True)
            n.src = [src]
        elif typ in ("ReLU", "PReLU", 18):
            src = blobs[bottoms[0]]
            if src.kind != "conv" or src.act != _ffi.PH_ACT_NONE or src.consumers:
                raise ValueError(f"layer {name!r}: activation must directly follow a convolution")
            src.act = _ffi.PH_ACT_PRELU if typ == "PReLU" else _ffi.PH_ACT_RELU
            src.act_name = name
            blobs[tops[0]] = src
            last = src
            continue
        elif typ in ("Pooling", 17):
            p = l["pooling_param"][0]
            if _one(p, "pool", "MAX") not in ("MAX", 0) or _one(p, "kernel_size") != 2 or _one(p, "stride") != 2 \
                    or _one(p, "pad", 0) != 0:
                raise ValueError(f"layer {name!r}: only 2x2 stride-2 MAX pooling is supported")
            src = blobs[bottoms[0]]
            n = _Node("pool", name, src.cout, (src.h + 1) // 2, (src.w + 1) // 2)
            n.src = [src]
        elif typ in ("Concat", 3):
            p = l.get("concat_param", [{}])[0]
            if _one(p, "axis", 1) != 1 or _one(p, "concat_dim", 1) != 1:
                raise ValueError(f"layer {name!r}: only channel concat is supported")
            srcs = [blobs[b] for b in bottoms]
            if any(s.kind == "concat" for s in srcs):
                raise ValueError(f"layer {name!r}: nested concat is not supported")
            n = _Node("concat", name, sum(s.cout for s in srcs), srcs[0].h, srcs[0].w)
            n.src = srcs
        else:
            raise ValueError(f"layer {name!r}: unsupported layer type {typ!r}")
        for s in n.src:
            s.consumers.append(n)
        blobs[tops[0]] = n
        nodes.append(n)
        last = n
    return nodes, inp, last


def build_plan(net, weights, height=None, width=None, tile_overrides=None, fuse_pool=True):
    """net: parsed prototxt; weights: {layer name: [blobs]}."""
    nodes, inp, out = _build_nodes(net, height, width)
    plan = Plan()
    plan.in_h, plan.in_w = inp.h, inp.w
    plan.out_channels, plan.out_h, plan.out_w = out.cout, out.h, out.w
    if inp.cout != 3:
        raise ValueError("network input must have 3 channels (BGR)")

    def new_buf(channels, h, w):
        plan.bufs.append(dict(channels=_r8(channels), height=h, width=w, pad=0))
        return len(plan.bufs) - 1

    # buffer 0: the preprocessed frame (3 -> 8 channels)
    inp.views = [(new_buf(8, inp.h, inp.w), 0)]

    # network output: fp32 windows
    if out.kind == "concat":
        off = 0
        for s in out.src:
            if s.kind != "conv":
                raise ValueError("network output concat members must be convolutions")
            s.f32_coff = off
            off += s.cout
    elif out.kind == "conv":
        out.f32_coff = 0
    else:
        raise ValueError("network output must be a convolution or a concat of convolutions")

    # concat buffers (those read by something)
    concat_layout = {}
    for n in nodes:
        if n.kind == "concat" and n is not out:
            if not n.consumers:
                continue
            # whole-buffer reads in 32-channel chunks (3x3 halo-patch kernel): pad the tail
            b = new_buf((sum(_r8(s.cout) for s in n.src) + 31) // 32 * 32, n.h, n.w)
            off = 0
            lay = []
            for s in n.src:
                s.views.append((b, off))
                lay.append((off, s.cout))
                off += _r8(s.cout)
            concat_layout[n.name] = (b, lay)
    # conv -> 2x2 max pool with nothing else reading the conv: the pool moves into the conv epilogue
    # (the full-resolution blob is never written)
    for n in nodes:
        n.fused_pool = None
    for n in nodes:
        if n.kind == "pool" and fuse_pool and not os.environ.get("POSEHIP_NO_POOL_FUSION"):
            c = n.src[0]
            if (c.kind == "conv" and c.k == 3 and len(c.consumers) == 1 and not c.views and c.f32_coff < 0
                    and c.h % 2 == 0 and c.w % 2 == 0 and _r8(c.src[0].cout) % 32 == 0
                    and c.src[0].kind != "concat"):
                c.fused_pool = n
                n.kind = "fused_pool"
    # own buffers for blobs read directly that have no view yet
    for n in nodes:
        if n.kind in ("conv", "pool", "fused_pool") and not n.views:
            if any(c.kind in ("conv", "pool") for c in n.consumers):
                n.views = [(new_buf(n.cout, n.h, n.w), 0)]

    def read_view(src, consumer):
        """(buf, coff, cin_read, cin_map) for `consumer` reading blob `src`."""
        if src.kind == "concat":
            b, lay = concat_layout[src.name]
            total = plan.bufs[b]["channels"]
            cmap = -np.ones(total, np.int32)
            caffe = 0
            for off, c in lay:
                cmap[off:off + c] = np.arange(caffe, caffe + c)
                caffe += c
            return b, 0, total, cmap
        b, off = src.views[0]
        cr = _r8(src.cout)
        cmap = -np.ones(cr, np.int32)
        cmap[:src.cout] = np.arange(src.cout)
        return b, off, cr, cmap

    for n in nodes:
        if n.kind == "conv":
            src = n.src[0]
            b, coff, cr, cmap = read_view(src, n)
            plan.bufs[b]["pad"] = max(plan.bufs[b]["pad"], (n.k - 1) // 2)
            if n.name not in weights:
                raise ValueError(f"no weights for layer {n.name!r}")
            blobs = weights[n.name]
            w = np.asarray(blobs[0], np.float32).reshape(n.cout, -1, n.k, n.k)
            if w.shape[1] != src.cout:
                raise ValueError(f"layer {n.name!r}: weight has {w.shape[1]} input channels, blob has {src.cout}")
            bias = np.asarray(blobs[1], np.float32).reshape(-1) if n.bias_term and len(blobs) > 1 else None
            slope = None
            if n.act == _ffi.PH_ACT_PRELU:
                if n.act_name not in weights:
                    raise ValueError(f"no PReLU slopes for layer {n.act_name!r}")
                slope = np.asarray(weights[n.act_name][0], np.float32).reshape(-1)
            views = n.fused_pool.views if n.fused_pool is not None else n.views
            if len(views) > _ffi.PH_MAX_DST:
                raise ValueError(f"layer {n.name!r} feeds {len(views)} buffers (max {_ffi.PH_MAX_DST})")
            op = dict(type=_ffi.PH_OP_CONV, name=n.name, in_buf=b, in_coff=coff, cin_read=cr, k=n.k,
                      cout=n.cout, act=n.act, dst=list(views), f32_coff=n.f32_coff, weight=w, bias=bias,
                      slope=slope, cin_map=cmap, h=n.h, w=n.w, pool=int(n.fused_pool is not None))
            if tile_overrides and n.name in tile_overrides:
                op["tile_n"] = tile_overrides[n.name]
            plan.ops.append(op)
            plan.op_names.append(n.name)
        elif n.kind == "pool":
            src = n.src[0]
            if not src.views:
                raise ValueError(f"layer {n.name!r}: pooling input {src.name!r} has no buffer of its own "
                                 f"(a Concat cannot be pooled)")
            b, coff = src.views[0]
            if coff != 0 or plan.bufs[b]["channels"] != _r8(src.cout) or len(n.views) != 1:
                # pooling reads/writes whole buffers; give the source its own
                raise ValueError(f"layer {n.name!r}: pooling input must own its buffer")
            plan.ops.append(dict(type=_ffi.PH_OP_POOL, name=n.name, in_buf=b, in_coff=0,
                                 cin_read=plan.bufs[b]["channels"], k=2, cout=n.cout, act=0,
                                 dst=list(n.views), f32_coff=-1, h=n.h, w=n.w))
            plan.op_names.append(n.name)
    return plan


def random_weights(net, seed=0, out_gain=0.05):
    """He-normal weights for every Convolution / PReLU of a parsed prototxt
    (no model files exist offline).  The convolutions feeding the
    network output are scaled by `out_gain` so heat maps stay O(0.1)."""
    rng = np.random.default_rng(seed)
    nodes, inp, out = _build_nodes(net, None, None)
    finals = set(s.name for s in out.src) if out.kind == "concat" else {out.name}
    w = {}
    for n in nodes:
        if n.kind != "conv":
            continue
        cin = n.src[0].cout
        std = np.sqrt(2.0 / (cin * n.k * n.k))
        if n.name in finals:
            std *= out_gain
        w[n.name] = [rng.normal(0, std, (n.cout, cin, n.k, n.k)).astype(np.float32),
                     rng.normal(0, 0.01, (n.cout,)).astype(np.float32)]
        if n.act == _ffi.PH_ACT_PRELU:
            w[n.act_name] = [rng.uniform(0.1, 0.3, (n.cout,)).astype(np.float32)]
    return w
