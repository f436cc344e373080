# mirrors openpose/opencv/__init__.py:1-2 of the reference
from posehip.api_opencv import PyOpenPoseOpenCV, get_body_parts_and_pose_pairs  # noqa: F401
