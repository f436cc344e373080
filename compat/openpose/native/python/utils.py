# openpose/native/python/utils.py
from posehip.utils import (Error, Timer, cosine_distance, dict_check, get_color, get_color_fast,  # noqa: F401
                           resize_tensor, str2bool)
