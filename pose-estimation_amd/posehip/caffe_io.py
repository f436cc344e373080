"""Caffe model files without Caffe: prototxt text parser, caffemodel (protobuf
wire format) reader/writer, and generators for the OpenPose BODY_25 / COCO
deploy graphs.

Stands in for ``cv2.dnn.readNetFromCaffe(proto_file, weights_file)``
(``openpose/opencv/skeleton.py:50``) and OpenPose's own Caffe loader behind
``model_folder`` / ``model_pose`` (``openpose/native/python/skeleton.py:386-387``).
No model files exist offline, so :func:`body25_prototxt` / :func:`coco_prototxt`
emit the published architectures (restated from the upstream deploy files,
unverified against them) and :func:`random_weights` fills them with He-normal
weights; real ``.prototxt`` / ``.caffemodel`` pairs load through the same path.
"""
import struct

import numpy as np

# ---------------------------------------------------------------------------
# prototxt (protobuf text format) -> nested dicts; every field is a list
# ---------------------------------------------------------------------------


def _tokenize(text):
    i, n = 0, len(text)
    while i < n:
        ch = text[i]
        if ch in " \t\r\n,;":
            i += 1
        elif ch == "#":
            while i < n and text[i] != "\n":
                i += 1
        elif ch in "{}:<>":
            yield ch
            i += 1
        elif ch in "\"'":
            j = i + 1
            while j < n and text[j] != ch:
                j += 2 if text[j] == "\\" else 1
            yield ("str", text[i + 1:j])
            i = j + 1
        else:
            j = i
            while j < n and text[j] not in " \t\r\n{}:<>#,;\"'":
                j += 1
            yield ("tok", text[i:j])
            i = j


def _scalar(tok):
    kind, val = tok
    if kind == "str":
        return val
    try:
        return int(val)
    except ValueError:
        pass
    try:
        return float(val)
    except ValueError:
        pass
    if val == "true":
        return True
    if val == "false":
        return False
    return val  # enum identifier


def parse_prototxt(text):
    toks = list(_tokenize(text))
    pos = 0

    def message(end):
        nonlocal pos
        msg = {}
        while pos < len(toks):
            t = toks[pos]
            if t in ("}", ">"):
                if end is None:
                    raise ValueError("unbalanced '}' in prototxt")
                pos += 1
                return msg
            if not isinstance(t, tuple):
                raise ValueError(f"unexpected token {t!r} in prototxt")
            name = t[1]
            pos += 1
            if pos >= len(toks):
                raise ValueError(f"field {name!r} has no value")
            if toks[pos] == ":":
                pos += 1
                if toks[pos] in ("{", "<"):
                    pos += 1
                    val = message("}")
                else:
                    val = _scalar(toks[pos])
                    pos += 1
            elif toks[pos] in ("{", "<"):
                pos += 1
                val = message("}")
            else:
                raise ValueError(f"field {name!r}: expected ':' or '{{'")
            msg.setdefault(name, []).append(val)
        if end is not None:
            raise ValueError("unterminated message in prototxt")
        return msg

    return message(None)


# ---------------------------------------------------------------------------
# caffemodel wire format
# ---------------------------------------------------------------------------


def _varint(buf, pos):
    res = shift = 0
    while True:
        b = buf[pos]
        pos += 1
        res |= (b & 0x7f) << shift
        if not b & 0x80:
            return res, pos
        shift += 7


def _fields(buf):
    """Yield (field_number, wire_type, value) over one message."""
    pos, n = 0, len(buf)
    while pos < n:
        key, pos = _varint(buf, pos)
        fn, wt = key >> 3, key & 7
        if wt == 0:
            v, pos = _varint(buf, pos)
        elif wt == 1:
            v = buf[pos:pos + 8]
            pos += 8
        elif wt == 2:
            ln, pos = _varint(buf, pos)
            v = buf[pos:pos + ln]
            pos += ln
        elif wt == 5:
            v = buf[pos:pos + 4]
            pos += 4
        else:
            raise ValueError(f"unsupported protobuf wire type {wt}")
        yield fn, wt, v


def _blob(buf):
    dims = {}
    shape = None
    data = []
    for fn, wt, v in _fields(buf):
        if fn in (1, 2, 3, 4) and wt == 0:      # num, channels, height, width
            dims[fn] = v
        elif fn == 5:                            # data (packed or repeated float)
            data.append(np.frombuffer(bytes(v), dtype="<f4"))
        elif fn == 8:                            # double_data
            data.append(np.frombuffer(bytes(v), dtype="<f8").astype(np.float32))
        elif fn == 7:                            # BlobShape { repeated int64 dim = 1 }
            shape = []
            for f2, w2, v2 in _fields(v):
                if f2 == 1 and w2 == 2:
                    p = 0
                    while p < len(v2):
                        d, p = _varint(v2, p)
                        shape.append(d)
                elif f2 == 1:
                    shape.append(v2)
    arr = np.concatenate(data) if data else np.zeros((0,), np.float32)
    if shape is None:
        shape = [dims.get(k, 1) for k in (1, 2, 3, 4)] if dims else [arr.size]
    return arr.astype(np.float32).reshape(shape)


def read_caffemodel(path):
    """-> {layer name: [np.float32 blobs]} (V2 `layer` = 100 and V1 `layers` = 2)."""
    with open(path, "rb") as f:
        buf = memoryview(f.read())
    out = {}
    for fn, wt, v in _fields(buf):
        if wt != 2 or fn not in (100, 2):
            continue
        name_f, blobs_f = (1, 7) if fn == 100 else (4, 6)
        name, blobs = None, []
        for f2, w2, v2 in _fields(v):
            if f2 == name_f and w2 == 2:
                name = bytes(v2).decode()
            elif f2 == blobs_f and w2 == 2:
                blobs.append(_blob(v2))
        if name is not None and blobs:
            out[name] = blobs
    return out


def _enc_varint(v):
    out = bytearray()
    while True:
        b = v & 0x7f
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _enc_ld(fn, payload):
    return _enc_varint((fn << 3) | 2) + _enc_varint(len(payload)) + payload


def write_caffemodel(path, layers):
    """layers: {name: [blobs]} -> NetParameter with V2 `layer` entries."""
    with open(path, "wb") as f:
        for name, blobs in layers.items():
            body = _enc_ld(1, name.encode()) + _enc_ld(2, b"Convolution")
            for b in blobs:
                b = np.ascontiguousarray(b, dtype="<f4")
                dims = b"".join(_enc_varint(d) for d in b.shape)
                shape = _enc_ld(1, dims)
                body += _enc_ld(7, _enc_ld(7, shape) + _enc_ld(5, b.tobytes()))
            f.write(_enc_ld(100, body))


# ---------------------------------------------------------------------------
# OpenPose deploy graphs (restated)
# ---------------------------------------------------------------------------


def _conv(name, bottom, cout, k, top=None):
    pad = (k - 1) // 2
    return (f'layer {{ name: "{name}" type: "Convolution" bottom: "{bottom}" top: "{top or name}"\n'
            f'  convolution_param {{ num_output: {cout} pad: {pad} kernel_size: {k} }} }}\n')


def _act(kind, name, blob):
    return f'layer {{ name: "{name}" type: "{kind}" bottom: "{blob}" top: "{blob}" }}\n'


def _pool(name, bottom):
    return (f'layer {{ name: "{name}" type: "Pooling" bottom: "{bottom}" top: "{name}"\n'
            f'  pooling_param {{ pool: MAX kernel_size: 2 stride: 2 }} }}\n')


def _concat(name, bottoms):
    bs = " ".join(f'bottom: "{b}"' for b in bottoms)
    return f'layer {{ name: "{name}" type: "Concat" {bs} top: "{name}" concat_param {{ axis: 1 }} }}\n'


def _vgg_front(act_kind_cpm):
    s = ""
    prev = "image"
    cfg = [("conv1_1", 64), ("conv1_2", 64), "pool1_stage1", ("conv2_1", 128), ("conv2_2", 128),
           "pool2_stage1", ("conv3_1", 256), ("conv3_2", 256), ("conv3_3", 256), ("conv3_4", 256),
           "pool3_stage1", ("conv4_1", 512), ("conv4_2", 512)]
    for item in cfg:
        if isinstance(item, str):
            s += _pool(item, prev)
            prev = item
        else:
            name, c = item
            s += _conv(name, prev, c, 3)
            s += _act("ReLU", "relu" + name[4:], name)
            prev = name
    s += _conv("conv4_3_CPM", prev, 256, 3)
    s += _act(act_kind_cpm, "prelu4_3_CPM" if act_kind_cpm == "PReLU" else "relu4_3_CPM", "conv4_3_CPM")
    s += _conv("conv4_4_CPM", "conv4_3_CPM", 128, 3)
    s += _act(act_kind_cpm, "prelu4_4_CPM" if act_kind_cpm == "PReLU" else "relu4_4_CPM", "conv4_4_CPM")
    return s


def body25_prototxt(height=368, width=656):
    """OpenPose BODY_25 ``pose_deploy.prototxt``: VGG-19 front (10 convs) +
    conv4_3/4_4_CPM (PReLU) + 4 PAF stages (52 ch) + 2 heat-map stages (26 ch),
    each stage = 5 dense blocks of 3x(3x3 conv) + 1x1 + 1x1; output
    ``net_output`` = concat(heat 26, PAF 52)."""
    s = f'name: "OpenPose - BODY_25"\ninput: "image"\ninput_dim: 1\ninput_dim: 3\ninput_dim: {height}\ninput_dim: {width}\n'
    s += _vgg_front("PReLU")

    def stage(tag, bottom, width_c, mid, cout):
        out = ""
        prev = bottom
        for blk in range(1, 6):
            names = [f"Mconv{blk}_stage{tag}_{i}" for i in range(3)]
            src = prev
            for nm in names:
                out += _conv(nm, src, width_c, 3)
                out += _act("PReLU", "Mprelu" + nm[5:], nm)
                src = nm
            cat = f"Mconv{blk}_stage{tag}_concat"
            out += _concat(cat, names)
            prev = cat
        n6, n7 = f"Mconv6_stage{tag}", f"Mconv7_stage{tag}"
        out += _conv(n6, prev, mid, 1)
        out += _act("PReLU", "Mprelu" + n6[5:], n6)
        out += _conv(n7, n6, cout, 1)
        return out, n7

    feat = "conv4_4_CPM"
    t, l2 = stage("0_L2", feat, 96, 256, 52)
    s += t
    for k in (1, 2, 3):
        cat = f"concat_stage{k}_L2"
        s += _concat(cat, [feat, l2])
        t, l2 = stage(f"{k}_L2", cat, 128, 512, 52)
        s += t
    s += _concat("concat_stage0_L1", [feat, l2])
    t, l1 = stage("0_L1", "concat_stage0_L1", 96, 256, 26)
    s += t
    s += _concat("concat_stage1_L1", [feat, l1, l2])
    t, l1 = stage("1_L1", "concat_stage1_L1", 128, 512, 26)
    s += t
    s += _concat("net_output", [l1, l2])
    return s


def coco_prototxt(height=368, width=368, n_paf=38, n_heat=19):
    """OpenPose COCO ``pose_deploy_linevec.prototxt``: VGG-19 front + CPM
    (ReLU) + stage 1 (3x 3x3, 1x1 512, 1x1 out) + stages 2-6 (5x 7x7, 2x 1x1);
    output = concat(heat 19, PAF 38)."""
    s = f'name: "OpenPose - COCO"\ninput: "image"\ninput_dim: 1\ninput_dim: 3\ninput_dim: {height}\ninput_dim: {width}\n'
    s += _vgg_front("ReLU")
    feat = "conv4_4_CPM"
    outs = {}
    for L, cout in (("L1", n_paf), ("L2", n_heat)):
        prev = feat
        for i in (1, 2, 3):
            nm = f"conv5_{i}_CPM_{L}"
            s += _conv(nm, prev, 128, 3)
            s += _act("ReLU", f"relu5_{i}_CPM_{L}", nm)
            prev = nm
        nm = f"conv5_4_CPM_{L}"
        s += _conv(nm, prev, 512, 1)
        s += _act("ReLU", f"relu5_4_CPM_{L}", nm)
        s += _conv(f"conv5_5_CPM_{L}", nm, cout, 1)
        outs[L] = f"conv5_5_CPM_{L}"
    for st in range(2, 7):
        cat = f"concat_stage{st}"
        s += _concat(cat, [outs["L1"], outs["L2"], feat])
        for L, cout in (("L1", n_paf), ("L2", n_heat)):
            prev = cat
            for i in range(1, 6):
                nm = f"Mconv{i}_stage{st}_{L}"
                s += _conv(nm, prev, 128, 7)
                s += _act("ReLU", f"Mrelu{i}_stage{st}_{L}", nm)
                prev = nm
            nm = f"Mconv6_stage{st}_{L}"
            s += _conv(nm, prev, 128, 1)
            s += _act("ReLU", f"Mrelu6_stage{st}_{L}", nm)
            s += _conv(f"Mconv7_stage{st}_{L}", nm, cout, 1)
            outs[L] = f"Mconv7_stage{st}_{L}"
    s += _concat("concat_stage7", [outs["L2"], outs["L1"]])
    return s
