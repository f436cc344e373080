"""posehip -- MI355X-native OpenPose bottom-up inference (host side).

The compute lives in lib/libposehip.so (hand-written gfx950 HIP, C ABI in
include/posehip.h); this package mirrors the reference's Python estimator
interface on top of it.
"""
from .tables import (COLORS, MAPIDX, TORSO_IDS, get_body_parts_and_pose_pairs,  # noqa: F401
                     BODY_25, COCO_18, V2_MODELS)

__all__ = ["COLORS", "MAPIDX", "TORSO_IDS", "get_body_parts_and_pose_pairs", "BODY_25", "COCO_18", "V2_MODELS"]
