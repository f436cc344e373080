# openpose/native/python/utils_rs.py
from posehip.utils_rs import (get_rs_sensor_dir, prepare_save_paths, read_calib_file, read_color_file,  # noqa: F401
                              read_depth_file)
