"""ORACLE (test infrastructure only -- never imported by the product path).

NumPy restatement of the two in-tree post-processing steps that follow the V2 decode:

* ``torso_bbox``  -- ``PyOpenPoseNativeBase.pose_bounding_box``, openpose/native/python/skeleton.py:89-113
* ``lift_3d``     -- ``PyOpenPoseNativeBase.get_3d_skeleton``,   openpose/native/python/skeleton.py:130-160

Both are plain NumPy in the reference, so this is one of the few places with an exact in-tree oracle.  One
numeric choice is made explicit: the lift is evaluated in float64 (keypoints converted first), which is what the
reference computes under NumPy < 2 value-based promotion (float32 scalar - Python float -> float64); under
NumPy >= 2 (NEP 50) the reference's ``x - cx`` would round to float32 first.  Pinned by hand-computed cases in
tests/test_oracle_v2.py.
"""
import numpy as np


def torso_bbox(pose_keypoints, torso_ids):
    """[M,J,3] f32 -> [M,4] f32 (u_min, v_min, u_max, v_max)."""
    bb = []
    for kp in np.asarray(pose_keypoints, np.float32):
        u, v, s = kp[torso_ids, 0], kp[torso_ids, 1], kp[torso_ids, 2]
        if (s != 0).any():
            u_min, u_max = u[s != 0].min(), u[s != 0].max()
            v_min, v_max = v[s != 0].min(), v[s != 0].max()
            if u_min == u_max:
                u_max = np.float32(u_max + np.float32(1))
            if v_min == v_max:
                v_max = np.float32(v_max + np.float32(1))
        else:
            u_min = v_min = np.float32(0)
            u_max = v_max = np.float32(1)
        bb.append(np.asarray([u_min, v_min, u_max, v_max], np.float32))
    return np.stack(bb) if bb else np.zeros((0, 4), np.float32)


def lift_3d(skeleton, depth_img, fx, fy, cx, cy, depth_scale=1e-3, patch_offset=2):
    """[J,3] f32 + u16 depth [H,W] -> [J,3] f64."""
    H, W = depth_img.shape
    out = []
    for x, y, _ in np.asarray(skeleton, np.float32):
        x, y = float(x), float(y)
        patch = depth_img[max(0, int(y - patch_offset)):min(H, int(y + patch_offset)),
                          max(0, int(x - patch_offset)):min(W, int(x + patch_offset))]
        avg = float(patch.astype(np.float64).sum() / patch.size) if patch.size else float("nan")
        x3d = (x - cx) / fx * avg
        y3d = (y - cy) / fy * avg
        out.append([x3d * depth_scale, y3d * depth_scale, avg * depth_scale])
    return np.array(out, np.float64).reshape(-1, 3)
