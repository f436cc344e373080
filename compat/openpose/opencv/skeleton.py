from posehip.api_opencv import PyOpenPoseOpenCV  # noqa: F401
