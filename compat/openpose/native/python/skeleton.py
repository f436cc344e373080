# mirrors the names callers import from openpose/native/python/skeleton.py (inference.py:12-13, tracking/track.py:8)
from posehip.api_native import (Datum, OpenPosePoseExtractor, PyOpenPoseNative, PyOpenPoseNativeBase,  # noqa: F401
                                TORSO_IDS)
