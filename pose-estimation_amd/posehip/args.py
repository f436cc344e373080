"""Command-line flags of the reference's pipeline callers
(``openpose/native/python/args.py:5-182``): every ``--op-*`` / tracker flag with the
reference's name, type and default, so ``get_parser().parse_known_args()`` yields a namespace
``OpenPosePoseExtractor`` and the callers in :mod:`posehip.inference` read unchanged.

Additions (``--posehip-*``): what the MI355X engine needs and the reference has no flag for.
"""
import argparse

from .utils import str2bool

# (flag, type, default, help)
_OPENPOSE = [
    ('--op-mode', str, "online", 'what mode to run the inference. [single,online,offline,comparetracker]'),
    ('--op-model-folder', str, "/usr/local/src/openpose/models/", 'folder with trained openpose models.'),
    ('--op-model-pose', str, "BODY_25", 'pose model name'),
    ('--op-net-resolution', str, "-1x368", 'resolution of input to openpose.'),
    ('--op-skel-thres', float, 0.5, 'threshold for valid skeleton.'),
    ('--op-max-true-body', int, 8, 'max number of skeletons to save.'),
    ('--op-heatmaps-add-parts', str2bool, False, ''),
    ('--op-heatmaps-add-bkg', str2bool, False, ''),
    ('--op-heatmaps-add-PAFs', str2bool, False, ''),
    ('--op-heatmaps-scale', int, 1, ''),
    ('--op-save-skel-folder', str, "", 'folder to save skeleton results.'),
    ('--op-save-skel', str2bool, False, 'if true, saves the 2d skeleton.'),
    ('--op-save-skel-image', str2bool, False, 'if true, saves the 2d skeleton image.'),
    ('--op-input-color-image', str, "", 'path to input color image/folder.'),
    ('--op-image-width', int, 1920, 'image width in px'),
    ('--op-image-height', int, 1080, 'image height in px'),
]
_DEPTH = [
    ('--op-patch-offset', int, 2, 'offset of patch used to determine depth'),
    ('--op-extract-3d-skel', str2bool, False, 'if true, tries to extract 3d skeleton.'),
    ('--op-save-3d-skel', str2bool, False, 'if true, saves the 3d skeleton.'),
]
_DISPLAY = [
    ('--op-display', float, 1.0, 'scale for displaying skel images.'),
    ('--op-display-depth', int, 0, 'scale for displaying skel images with depth.'),
]
_REALSENSE = [
    ('--op-rs-dir', str, 'data/mot17', 'path to folder with saved rs data.'),
    ('--op-rs-delete-image', str2bool, False, 'if true, deletes the rs image used in inference.'),
]
_TRACKING = [
    ('--op-track-deepsort', str2bool, False, 'If true performs deepsort tracking.'),
    ('--op-track-bytetrack', str2bool, False, 'If true performs ByteTrack tracking.'),
    ('--op-track-ocsort', str2bool, False, 'If true performs OC Sort tracking.'),
    ('--op-track-strongsort', str2bool, False, 'If true performs StrongSort tracking.'),
    ('--op-track-buffer', int, 30, 'the frames for keep lost tracks.'),
    ('--deepsort-metric', str, 'cosine', None),
    ('--deepsort-opt-nsa', str2bool, False, None),
    ('--deepsort-opt-ema', str2bool, False, None),
    ('--deepsort-opt-emaalpha', float, 0.0, None),
    ('--deepsort-opt-mc', str2bool, False, None),
    ('--deepsort-opt-mclambda', float, 0.0, None),
    ('--deepsort-opt-woc', str2bool, False, None),
    ('--deepsort-opt-maxcosinedistance', float, 0.2, None),
    ('--deepsort-opt-nnbudget', int, 100, None),
    ('--strongsort-metric', str, 'cosine', None),
    ('--strongsort-opt-nsa', str2bool, True, None),
    ('--strongsort-opt-ema', str2bool, True, None),
    ('--strongsort-opt-emaalpha', float, 0.9, None),
    ('--strongsort-opt-mc', str2bool, True, None),
    ('--strongsort-opt-mclambda', float, 0.98, None),
    ('--strongsort-opt-woc', str2bool, True, None),
    ('--bytetracker-trackthresh', float, 0.5, None),
    ('--bytetracker-trackbuffer', int, 30, None),
    ('--bytetracker-matchthresh', float, 0.8, None),
    ('--bytetracker-mot20', str2bool, False, None),
    ('--bytetracker-fps', float, 30, None),
    ('--ocsort-detthresh', float, 0.5, None),
    ('--ocsort-maxage', int, 30, None),
    ('--ocsort-minhits', int, 3, None),
    ('--ocsort-iouthreshold', float, 0.3, None),
    ('--ocsort-deltat', int, 3, None),
    ('--ocsort-assofunc', str, "iou", None),
    ('--ocsort-inertia', float, 0.2, None),
    ('--ocsort-usebyte', str2bool, True, None),
]
_EXPERIMENTS = [
    ('--op-save-track-image', str2bool, False, 'If true saves tracking+pose results in image form.'),
    ('--op-proc', str, 'sp', 'multi (mp) or single processing (sp) or none. Based on Alphapose'),
]
_POSEHIP = [
    ('--posehip-random-weights', int, None, 'seed: run the graph with random weights when no model files exist'),
    ('--posehip-device', int, None, 'GPU index (default 0)'),
    ('--posehip-batch', int, None, 'frames per engine call in the batched offline mode (default 1)'),
]

REFERENCE_FLAGS = _OPENPOSE + _DEPTH + _DISPLAY + _REALSENSE + _TRACKING + _EXPERIMENTS


def get_parser() -> argparse.ArgumentParser:
    p = argparse.ArgumentParser(description='Extract skeleton using OPENPOSE')
    for flag, typ, default, text in REFERENCE_FLAGS + _POSEHIP:
        p.add_argument(flag, type=typ, default=default, help=text)
    return p
