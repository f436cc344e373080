"""Thin host wrapper over one ``ph_ctx`` (one device, one HIP stream).

All arithmetic happens in libposehip.so; this class only marshals NumPy / torch
buffers across the C ABI.
"""
import ctypes as C

import numpy as np

from . import _ffi
from .tables import V2_MODELS


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class Engine:
    def __init__(self, device=0, stream=None):
        self._lib = _ffi.lib()
        self._ctx = C.c_void_p()
        _ffi.check(self._lib.ph_create(C.byref(self._ctx), int(device),
                                       C.c_void_p(stream) if stream else None), "ph_create")
        self.device = device
        self.plan = None
        self.batch = 0
        self._v2_structs = {}

    def close(self):
        if getattr(self, "_ctx", None) and self._ctx.value:
            self._v1_release()
            self._lib.ph_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- network ---------------------------------------------------------
    def build(self, plan, batch):
        bufs, ops = plan.to_ctypes()
        _ffi.check(self._lib.ph_net_build(self._ctx, int(batch), bufs, len(plan.bufs), ops, len(plan.ops),
                                          plan.out_channels, plan.out_h, plan.out_w), "ph_net_build")
        self.plan = plan
        self.batch = batch

    def preprocess(self, frames, mode=_ffi.PH_PRE_CV2):
        """frames: u8 [B,H0,W0,3] BGR -- NumPy (host) or a torch CUDA tensor."""
        if isinstance(frames, np.ndarray):
            f = np.ascontiguousarray(frames, dtype=np.uint8)
            if f.ndim != 4 or f.shape[3] != 3:
                raise ValueError(f"frames must be u8 [B,H,W,3], got {f.shape}")
            B, H0, W0, _ = f.shape
            _ffi.check(self._lib.ph_preprocess(self._ctx, _ptr(f), 0, B, H0, W0, mode), "ph_preprocess")
            self._lib.ph_sync(self._ctx)  # host buffer may go away after return
        else:  # torch tensor on the device
            if frames.dim() != 4 or frames.shape[3] != 3 or not frames.is_contiguous() or not frames.is_cuda:
                raise ValueError("device frames must be a contiguous CUDA u8 [B,H,W,3] tensor")
            B, H0, W0, _ = frames.shape
            _ffi.check(self._lib.ph_preprocess(self._ctx, C.c_void_p(frames.data_ptr()), 1, B, H0, W0, mode),
                       "ph_preprocess")
        return B

    def forward(self, B):
        _ffi.check(self._lib.ph_forward(self._ctx, int(B)), "ph_forward")

    def forward_timed(self, B, wait=True):
        """Forward with per-op HIP events; wait=False only records them."""
        if not wait:
            _ffi.check(self._lib.ph_forward_timed(self._ctx, int(B), None), "ph_forward_timed")
            return None
        ms = (C.c_float * len(self.plan.ops))()
        _ffi.check(self._lib.ph_forward_timed(self._ctx, int(B), ms), "ph_forward_timed")
        return np.array(ms[:], dtype=np.float64)

    def forward_times(self):
        """Per-op device ms of the last forward_timed (synchronises the stream)."""
        ms = (C.c_float * len(self.plan.ops))()
        _ffi.check(self._lib.ph_forward_times(self._ctx, ms), "ph_forward_times")
        return np.array(ms[:], dtype=np.float64)

    def op_flops(self):
        return np.array([max(0.0, self._lib.ph_op_flops(self._ctx, i)) for i in range(len(self.plan.ops))])

    def op_kernels(self):
        """Kernel instantiation name of every op (rocprofv3 kernel-trace spelling)."""
        out = []
        buf = C.create_string_buffer(128)
        for i in range(len(self.plan.ops)):
            _ffi.check(self._lib.ph_op_kernel(self._ctx, i, buf, 128), "ph_op_kernel")
            out.append(buf.value.decode())
        return out

    def net_out(self, B):
        p = self.plan
        out = np.empty((B, p.out_channels, p.out_h, p.out_w), np.float32)
        _ffi.check(self._lib.ph_get_net_out(self._ctx, int(B), _ptr(out)), "ph_get_net_out")
        return out

    def set_net_out(self, arr):
        """Inject a net output, f32 [B,C,Hs,Ws] (NumPy or torch CUDA tensor)."""
        if isinstance(arr, np.ndarray):
            a = np.ascontiguousarray(arr, dtype=np.float32)
            B, Cc, Hs, Ws = a.shape
            _ffi.check(self._lib.ph_set_net_out(self._ctx, _ptr(a), 0, B, Cc, Hs, Ws), "ph_set_net_out")
        else:
            B, Cc, Hs, Ws = arr.shape
            _ffi.check(self._lib.ph_set_net_out(self._ctx, C.c_void_p(arr.data_ptr()), 1, B, Cc, Hs, Ws),
                       "ph_set_net_out")
        self._injected = (B, Cc, Hs, Ws)

    def upsample_maps(self, frame, c0, nc, Ht, Wt, n_heat=0, scale_mode=3):
        """Bicubic-upsampled maps [nc,Ht,Wt] of channels [c0, c0+nc) of one frame; `scale_mode` = OpenPose's
        --heatmaps_scale (3 = as is), channels below `n_heat` are part/background maps, the rest PAFs."""
        out = np.empty((nc, Ht, Wt), np.float32)
        _ffi.check(self._lib.ph_upsample_maps(self._ctx, int(frame), int(c0), int(nc), int(Ht), int(Wt), int(n_heat),
                                              int(scale_mode), _ptr(out)), "ph_upsample_maps")
        return out

    def argmax_maps(self, B, C_, Ws):
        """-> (max value f32 [B,C], x i32 [B,C], y i32 [B,C]) of the net output maps."""
        vals = np.empty((B * C_,), np.float32)
        locs = np.empty((B * C_,), np.int32)
        _ffi.check(self._lib.ph_argmax_maps(self._ctx, int(B), _ptr(vals), _ptr(locs)), "ph_argmax_maps")
        return vals.reshape(B, C_), (locs % Ws).reshape(B, C_), (locs // Ws).reshape(B, C_)

    def get_buffer(self, buf):
        """Debug: bf16 activation buffer as float32 [B,Hp,Wp,C]."""
        d = self.plan.bufs[buf]
        shape = (self.batch, d["height"] + 2 * d["pad"], d["width"] + 2 * d["pad"], d["channels"])
        raw = np.empty(shape, np.uint16)
        _ffi.check(self._lib.ph_get_buffer(self._ctx, buf, _ptr(raw), raw.size), "ph_get_buffer")
        return (raw.astype(np.uint32) << 16).view(np.float32)

    def sync(self):
        _ffi.check(self._lib.ph_sync(self._ctx), "ph_sync")

    # ---- decode ----------------------------------------------------------
    def _v2(self, model):
        if isinstance(model, str):
            model = V2_MODELS[model]
        key = id(model)
        if key not in self._v2_structs:
            self._v2_structs[key] = (_ffi.v2_model_struct(model), model)
        return self._v2_structs[key]

    def decode_v2(self, model, B, Ht, Wt, scale=1.0, max_people=128, want_peaks=False, want_subsets=False):
        """-> list of (keypoints f32 [M,J,3], scores f32 [M]) per frame (+extras)."""
        ms, model = self._v2(model)
        J = model.n_parts
        n = np.zeros((B,), np.int32)
        kp = np.zeros((B, max_people, J, 3), np.float32)
        sc = np.zeros((B, max_people), np.float32)
        peaks = np.zeros((B, J, model.max_peaks + 1, 3), np.float32) if want_peaks else None
        subs = np.zeros((B, max_people, J + 1), np.int32) if want_subsets else None
        _ffi.check(self._lib.ph_decode_v2(self._ctx, C.byref(ms), int(B), int(Ht), int(Wt), float(scale),
                                          int(max_people), _ptr(n), _ptr(kp), _ptr(sc),
                                          _ptr(peaks) if want_peaks else None,
                                          _ptr(subs) if want_subsets else None), "ph_decode_v2")
        res = []
        for b in range(B):
            m = min(int(n[b]), max_people)
            res.append((kp[b, :m].copy(), sc[b, :m].copy()))
        extras = dict(n_people=n, peaks=peaks, subsets=subs)
        return res, extras

    def decode_v2_dev(self, model, B, Ht, Wt, scale=1.0, max_people=128):
        """Decode and leave results on the device; returns raw device addresses
        (n_people i32[B], keypoints f32[B,max_people,J,3], scores f32[B,max_people])."""
        ms, model = self._v2(model)
        pn, pk, ps = C.c_void_p(), C.c_void_p(), C.c_void_p()
        _ffi.check(self._lib.ph_decode_v2_dev(self._ctx, C.byref(ms), int(B), int(Ht), int(Wt), float(scale),
                                              int(max_people), C.byref(pn), C.byref(pk), C.byref(ps)),
                   "ph_decode_v2_dev")
        return pn.value, pk.value, ps.value

    def decode_v2_results(self, n_people, keypoints, scores):
        """Async copy of the last decode_v2_dev results into torch tensors
        (pinned host or CUDA); the caller synchronises the stream."""
        on_dev = int(keypoints.is_cuda)
        _ffi.check(self._lib.ph_decode_v2_results(self._ctx, C.c_void_p(n_people.data_ptr()),
                                                  C.c_void_p(keypoints.data_ptr()), C.c_void_p(scores.data_ptr()),
                                                  on_dev), "ph_decode_v2_results")

    def decode_v2_status(self):
        """Waits for the stream; raises PosehipError (-2) if the decode whose results were last copied by
        decode_v2_results dropped people (more than 2048 partial skeletons in a frame)."""
        _ffi.check(self._lib.ph_decode_v2_status(self._ctx), "ph_decode_v2_status")

    def decode_v2_noalloc(self, model, net_out, Ht, Wt, n_people, keypoints, scores, scale=1.0, overflow=None):
        """The allocation-free decode: net_out f32 [B,C,Hs,Ws], n_people i32 [B], keypoints f32 [B,M,J,3],
        scores f32 [B,M] (and overflow i32 [1]) are the caller's CUDA tensors; asynchronous on the engine's stream.
        At most PH_NOALLOC_MAX_FRAMES BODY_25 frames per call (static 20 MiB workspace per GPU)."""
        ms, model = self._v2(model)
        for t in (net_out, n_people, keypoints, scores):
            if not t.is_cuda or not t.is_contiguous():
                raise ValueError("decode_v2_noalloc takes contiguous CUDA tensors")
        B, Cc, Hs, Ws = net_out.shape
        M = int(keypoints.shape[1])
        if tuple(keypoints.shape) != (B, M, model.n_parts, 3) or tuple(scores.shape) != (B, M) or n_people.numel() != B:
            raise ValueError("decode_v2_noalloc: output tensor shapes do not match the batch / model")
        _ffi.check(self._lib.ph_decode_v2_noalloc(
            self._ctx, C.byref(ms), C.c_void_p(net_out.data_ptr()), int(B), int(Cc), int(Hs), int(Ws), int(Ht), int(Wt),
            float(scale), M, C.c_void_p(n_people.data_ptr()), C.c_void_p(keypoints.data_ptr()),
            C.c_void_p(scores.data_ptr()), C.c_void_p(overflow.data_ptr()) if overflow is not None else None),
            "ph_decode_v2_noalloc")

    def pose_post(self, B, max_people, n_parts, torso_ids=None, depth=None, intr=None, depth_scale=1e-3,
                  patch_offset=2):
        """Torso boxes and / or 3-D lift of the last decode_v2(_dev) results, computed on the device keypoints.
        depth: u16 [B,Hd,Wd] NumPy array or CUDA tensor; intr = (fx, fy, cx, cy).
        -> (bbox f32 [B,max_people,4] | None, kp3d f64 [B,max_people,n_parts,3] | None)"""
        bbox = kp3d = ids = None
        if torso_ids is not None:
            ids = np.ascontiguousarray(torso_ids, np.int32)
            bbox = np.zeros((B, max_people, 4), np.float32)
        dptr, on_dev, Hd, Wd = None, 0, 0, 0
        fx = fy = cx = cy = 0.0
        if depth is not None:
            if isinstance(depth, np.ndarray):
                depth = np.ascontiguousarray(depth, np.uint16)
                dptr = _ptr(depth)
            else:
                if not depth.is_cuda or not depth.is_contiguous() or depth.element_size() != 2:
                    raise ValueError("device depth must be a contiguous CUDA 16-bit tensor [B,H,W]")
                dptr, on_dev = C.c_void_p(depth.data_ptr()), 1
            if depth.ndim != 3 or depth.shape[0] < B:
                raise ValueError(f"depth must be [B>={B},H,W], got {tuple(depth.shape)}")
            Hd, Wd = int(depth.shape[1]), int(depth.shape[2])
            fx, fy, cx, cy = (float(v) for v in intr)
            kp3d = np.zeros((B, max_people, n_parts, 3), np.float64)
        _ffi.check(self._lib.ph_pose_post(self._ctx, int(B), int(max_people), int(n_parts),
                                          _ptr(ids) if ids is not None else None, len(ids) if ids is not None else 0,
                                          _ptr(bbox) if bbox is not None else None, dptr, on_dev, Hd, Wd, fx, fy, cx, cy,
                                          float(depth_scale), int(patch_offset),
                                          _ptr(kp3d) if kp3d is not None else None), "ph_pose_post")
        return bbox, kp3d

    def pose_features(self, B, max_people, n_parts, torso_ids, windows, n_heat, scale_mode, Ht, Wt, image_size,
                      n_rows, size=(25, 50)):
        """Tracker appearance features of the people of the last decode_v2(_dev), straight from the resident net
        output (create_detections' crop + resize_tensor, tracking/utils_track.py:45-57).
        windows: [(c0, nc), ...] channel windows of the net output; image_size = (width, height) of the frames;
        n_rows: people of all B frames (known from the decode); size = (x, y) of the resampled crop.
        -> (feat f32 [n_rows, y, x, C], status i32 [n_rows]: 1 ok / -1 degenerate crop)"""
        ids = np.ascontiguousarray(torso_ids, np.int32)
        win = np.ascontiguousarray(windows, np.int32).reshape(-1, 2)
        nch = int(win[:, 1].sum())
        fx, fy = int(size[0]), int(size[1])
        cap = max(int(n_rows), 1)
        feat = np.zeros((cap, fy, fx, nch), np.float32)
        status = np.zeros((cap,), np.int32)
        total = C.c_int32(0)
        _ffi.check(self._lib.ph_pose_features(self._ctx, int(B), int(max_people), int(n_parts), _ptr(ids), len(ids),
                                              _ptr(win), len(win), int(n_heat), int(scale_mode), int(Ht), int(Wt),
                                              int(image_size[0]), int(image_size[1]), fx, fy, cap, _ptr(feat),
                                              _ptr(status), C.byref(total)), "ph_pose_features")
        return feat[:total.value], status[:total.value]

    def decode_v2_connections(self, model, B):
        """Test/debug: per-limb connection lists of the last decode."""
        _, model = self._v2(model)
        n = np.zeros((B, model.n_pairs), np.int32)
        idx = np.zeros((B, model.n_pairs, model.max_peaks, 2), np.int32)
        sc = np.zeros((B, model.n_pairs, model.max_peaks), np.float32)
        _ffi.check(self._lib.ph_decode_v2_connections(self._ctx, _ptr(n), _ptr(idx), _ptr(sc)),
                   "ph_decode_v2_connections")
        return n, idx, sc

    def decode_v1(self, body_parts, pose_pairs, mapidx, B, W0, H0, conf_thres=0.1, kp_cap=256, person_cap=512):
        """V1 decode of the resident net output -> per frame (keypoints_list f64
        [K,3], personwise list of f64 rows, extras) exactly as postprocess_multi
        returns them (openpose/opencv/skeleton.py:154)."""
        m = _ffi.v1_model_struct(body_parts, pose_pairs, mapidx)
        P, L = m.n_parts, m.n_limbs
        n_kps, kps, n_pairs, pairs, n_persons, persons = self._v1_host(B, P, L, kp_cap, person_cap)
        _ffi.check(self._lib.ph_decode_v1(self._ctx, C.byref(m), int(B), int(W0), int(H0), float(conf_thres),
                                          int(kp_cap), int(person_cap), _ptr(n_kps), _ptr(kps), _ptr(n_pairs),
                                          _ptr(pairs), _ptr(n_persons), _ptr(persons)), "ph_decode_v1")
        # everything handed back is a copy: the staging arrays are written again by the next call
        live = np.arange(kp_cap)[None, None, :] < n_kps[:, :, None]
        out = []
        for b in range(B):
            klist = kps[b][live[b]].astype(np.float64)          # parts in order, each part's keypoints in order
            people = list(persons[b, :n_persons[b]].copy())
            valid = [pairs[b, k, :n_pairs[b, k]].copy() if n_pairs[b, k] >= 0 else [] for k in range(L)]
            invalid = [k for k in range(L) if n_pairs[b, k] < 0]
            out.append((klist, people, dict(n_kps=n_kps[b].copy(), valid_pairs=valid, invalid_pairs=invalid)))
        return out

    def _v1_host(self, B, P, L, kp_cap, person_cap):
        """Page-locked staging arrays (one ph_host_alloc region) for ph_decode_v1's results, kept per shape: the call copies ~10 MB of
        capacity-sized tables, which through fresh pageable NumPy arrays (allocation, zero fill, the runtime's own
        staging) cost more than the decode kernels at 20 people."""
        key = (B, P, L, kp_cap, person_cap)
        if getattr(self, "_v1_key", None) != key:
            self._v1_release()
            shapes = (((B, P), np.int32), ((B, P, kp_cap, 3), np.float32), ((B, L), np.int32),
                      ((B, L, kp_cap, 3), np.float64), ((B,), np.int32), ((B, person_cap, P + 1), np.float64))
            sizes = [(int(np.prod(sh)) * np.dtype(dt).itemsize + 255) // 256 * 256 for sh, dt in shapes]
            total = sum(sizes)
            ptr = self._lib.ph_host_alloc(total)
            if not ptr:
                _ffi.check(-1, "ph_host_alloc")
            region = np.frombuffer((C.c_uint8 * total).from_address(ptr), np.uint8)
            arrs, off = [], 0
            for (sh, dt), sz in zip(shapes, sizes):
                n = int(np.prod(sh)) * np.dtype(dt).itemsize
                arrs.append(region[off:off + n].view(dt).reshape(sh))
                off += sz
            self._v1_arrs, self._v1_key, self._v1_ptr = tuple(arrs), key, ptr
        return self._v1_arrs

    def _v1_release(self):
        if getattr(self, "_v1_ptr", None):
            self._v1_arrs = ()                     # (no view may outlive the memory)
            self._lib.ph_host_free(C.c_void_p(self._v1_ptr))
        self._v1_arrs, self._v1_key, self._v1_ptr = (), None, None

    def decode_v1_times(self):
        ms = (C.c_float * 3)()
        _ffi.check(self._lib.ph_decode_v1_times(self._ctx, ms), "ph_decode_v1_times")
        return list(ms)

    def decode_v1_walks(self):
        """(blobs walked by a wave, blobs walked by one group) of the last decode_v1."""
        a, b = C.c_int32(0), C.c_int32(0)
        _ffi.check(self._lib.ph_decode_v1_walks(self._ctx, C.byref(a), C.byref(b)), "ph_decode_v1_walks")
        return a.value, b.value

    def decode_v2_times(self):
        ms = (C.c_float * 3)()
        _ffi.check(self._lib.ph_decode_v2_times(self._ctx, ms), "ph_decode_v2_times")
        return list(ms)
