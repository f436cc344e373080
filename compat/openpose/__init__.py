"""Import-path shim: put `compat/` and `pose-estimation_amd/` on PYTHONPATH and the reference's
callers (`from openpose.opencv import ...`, `from openpose.native.python.skeleton import ...`,
`from openpose.native.python.inference import extract_2dskeletons`, `...args import get_parser`,
`...utils import Timer`, `...utils_rs import read_color_file`) get the posehip implementations instead of the
cv2.dnn / pyopenpose ones.  The reference's trackers (`tracking/`) and demos are not part of this repository."""
