"""ctypes binding of libposehip.so (C ABI declared in include/posehip.h).

There is no CPU fallback: if the HIP library is missing or cannot be loaded the
import of any compute entry point raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# POSEHIP_LIB: another build of the same library (A/B timing of kernel variants); there is no other fallback
LIB_PATH = os.environ.get("POSEHIP_LIB") or os.path.join(os.path.dirname(_HERE), "lib", "libposehip.so")

PH_MAX_DST = 8
PH_OP_CONV, PH_OP_POOL = 0, 1
PH_ACT_NONE, PH_ACT_RELU, PH_ACT_PRELU = 0, 1, 2
PH_PRE_CV2, PH_PRE_NATIVE = 0, 1
PH_NOALLOC_MAX_FRAMES = 8


class PhBufDesc(C.Structure):
    _fields_ = [("channels", C.c_int), ("height", C.c_int), ("width", C.c_int),
                ("pad", C.c_int)]


class PhOpDesc(C.Structure):
    _fields_ = [("type", C.c_int), ("in_buf", C.c_int), ("in_coff", C.c_int),
                ("cin_read", C.c_int), ("kh", C.c_int), ("kw", C.c_int),
                ("cout", C.c_int), ("act", C.c_int), ("n_dst", C.c_int),
                ("dst_buf", C.c_int * PH_MAX_DST),
                ("dst_coff", C.c_int * PH_MAX_DST),
                ("f32_coff", C.c_int), ("cin_caffe", C.c_int),
                ("slope_count", C.c_int), ("tile_n", C.c_int), ("pool", C.c_int),
                ("weight", C.POINTER(C.c_float)),
                ("bias", C.POINTER(C.c_float)),
                ("slope", C.POINTER(C.c_float)),
                ("cin_map", C.POINTER(C.c_int32))]


class PhV2Model(C.Structure):
    _fields_ = [("n_parts", C.c_int), ("n_heat", C.c_int), ("n_pairs", C.c_int),
                ("max_peaks", C.c_int), ("ear_pair0", C.c_int),
                ("ear_pair1", C.c_int), ("min_subset_cnt", C.c_int),
                ("nms_threshold", C.c_float), ("inter_threshold", C.c_float),
                ("inter_min_above_threshold", C.c_float),
                ("min_subset_score", C.c_float),
                ("pairs", C.c_int * 128), ("mapidx", C.c_int * 128), ("connector", C.c_int)]


class PhV1Model(C.Structure):
    _fields_ = [("n_parts", C.c_int), ("n_limbs", C.c_int), ("pair_parts", C.c_int * 64),
                ("mapidx", C.c_int * 64)]


# every symbol include/posehip.h declares: name -> (restype, argtypes)
_vp = C.c_void_p
_fp = C.POINTER(C.c_float)
_ip = C.POINTER(C.c_int32)
SYMBOLS = {
    "ph_create": (C.c_int, [C.POINTER(_vp), C.c_int, _vp]),
    "ph_destroy": (None, [_vp]),
    "ph_sync": (C.c_int, [_vp]),
    "ph_last_error": (C.c_char_p, []),
    "ph_device_ok": (C.c_int, [C.c_int]),
    "ph_host_alloc": (_vp, [C.c_int64]),
    "ph_host_free": (C.c_int, [_vp]),
    "ph_net_build": (C.c_int, [_vp, C.c_int, C.POINTER(PhBufDesc), C.c_int,
                               C.POINTER(PhOpDesc), C.c_int, C.c_int, C.c_int,
                               C.c_int]),
    "ph_preprocess": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int,
                                C.c_int]),
    "ph_forward": (C.c_int, [_vp, C.c_int]),
    "ph_forward_timed": (C.c_int, [_vp, C.c_int, _fp]),
    "ph_forward_times": (C.c_int, [_vp, _fp]),
    "ph_op_flops": (C.c_double, [_vp, C.c_int]),
    "ph_op_kernel": (C.c_int, [_vp, C.c_int, C.c_char_p, C.c_int]),
    "ph_pose_post": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, _vp, C.c_int, _vp, _vp, C.c_int, C.c_int, C.c_int,
                               C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int, _vp]),
    "ph_get_net_out": (C.c_int, [_vp, C.c_int, _vp]),
    "ph_net_out_dev": (_vp, [_vp]),
    "ph_set_net_out": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int,
                                 C.c_int]),
    "ph_upsample_maps": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _vp]),
    "ph_movenet_crop_resize": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_int, _vp, C.c_int, _vp, C.c_int]),
    "ph_movenet_update": (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int, C.c_int, _vp, C.c_int]),
    "ph_movenet_multipose_decode": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_float, _vp,
                                              C.c_int]),
    "ph_pose_features": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, _vp, C.c_int, _vp, C.c_int, C.c_int, C.c_int,
                                   C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp]),
    "ph_argmax_maps": (C.c_int, [_vp, C.c_int, _vp, _vp]),
    "ph_get_buffer": (C.c_int, [_vp, C.c_int, _vp, C.c_int64]),
    "ph_decode_v2": (C.c_int, [_vp, C.POINTER(PhV2Model), C.c_int, C.c_int,
                               C.c_int, C.c_float, C.c_int, _vp, _vp, _vp,
                               _vp, _vp]),
    "ph_decode_v2_dev": (C.c_int, [_vp, C.POINTER(PhV2Model), C.c_int, C.c_int,
                                   C.c_int, C.c_float, C.c_int,
                                   C.POINTER(_vp), C.POINTER(_vp),
                                   C.POINTER(_vp)]),
    "ph_decode_v2_results": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int]),
    "ph_decode_v2_status": (C.c_int, [_vp]),
    "ph_decode_v2_noalloc": (C.c_int, [_vp, C.POINTER(PhV2Model), _vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                       C.c_int, C.c_float, C.c_int, _vp, _vp, _vp, _vp]),
    "ph_decode_v2_connections": (C.c_int, [_vp, _vp, _vp, _vp]),
    "ph_decode_v1": (C.c_int, [_vp, C.POINTER(PhV1Model), C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int,
                               _vp, _vp, _vp, _vp, _vp, _vp]),
    "ph_decode_v1_times": (C.c_int, [_vp, _fp]),
    "ph_decode_v1_walks": (C.c_int, [_vp, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "ph_decode_v2_times": (C.c_int, [_vp, _fp]),
}

_lib = None


class PosehipError(RuntimeError):
    pass


def lib():
    """Load libposehip.so once; raises loudly when it is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise PosehipError(
                f"{LIB_PATH} not found: build it with `make -C "
                f"{os.path.dirname(os.path.dirname(LIB_PATH))}` (hipcc, gfx950). "
                "posehip has no CPU fallback.")
        try:
            # torch bundles its own libamdhip64 (same SONAME): load it first so that
            # libposehip.so binds to the SAME HIP runtime and can share torch's streams
            import torch  # noqa: F401
        except ImportError:
            pass
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(l, name)
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


def check(rc, what=""):
    if rc != 0:
        msg = lib().ph_last_error()
        raise PosehipError(f"{what} failed ({rc}): "
                           f"{msg.decode() if msg else 'unknown error'}")


def v2_model_struct(model):
    """posehip.tables.PoseModelV2 -> PhV2Model."""
    m = PhV2Model()
    m.n_parts = model.n_parts
    m.n_heat = model.n_heat
    m.n_pairs = model.n_pairs
    m.max_peaks = model.max_peaks
    m.ear_pair0, m.ear_pair1 = model.ear_pair_indices
    m.min_subset_cnt = model.min_subset_cnt
    m.nms_threshold = model.nms_threshold
    m.inter_threshold = model.inter_threshold
    m.inter_min_above_threshold = model.inter_min_above_threshold
    m.min_subset_score = model.min_subset_score
    m.connector = int(getattr(model, "connector", 1))   # tables.PoseModelV2's default: the CUDA build's route
    for k, (a, b) in enumerate(model.pairs):
        m.pairs[2 * k], m.pairs[2 * k + 1] = a, b
    for k, (a, b) in enumerate(model.mapidx):
        m.mapidx[2 * k], m.mapidx[2 * k + 1] = a, b
    return m


def v1_model_struct(body_parts, pose_pairs, mapidx):
    """The reference's body_parts dict / pose_pairs list / MAPIDX -> PhV1Model."""
    m = PhV1Model()
    m.n_parts = len(body_parts) - 1
    m.n_limbs = len(mapidx)
    if m.n_limbs > len(pose_pairs) or m.n_limbs > 32 or m.n_parts > 32:
        raise ValueError("V1 model tables too large / inconsistent")
    for k in range(m.n_limbs):
        m.pair_parts[2 * k] = body_parts[pose_pairs[k][0]]
        m.pair_parts[2 * k + 1] = body_parts[pose_pairs[k][1]]
        m.mapidx[2 * k], m.mapidx[2 * k + 1] = mapidx[k]
    return m
