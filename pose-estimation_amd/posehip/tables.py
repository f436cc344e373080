"""Skeleton tables for the OpenPose bottom-up decode.

COCO-18 / MPI-15 tables restate ``openpose/opencv/common.py:2-102`` of the
reference (COLORS :2-21, MAPIDX :26-28, get_body_parts_and_pose_pairs :31-102).

BODY_25 (and the full 19-limb COCO tables used by the upstream-semantics
"V2" decode) are NOT in the reference tree.  They are restated from the public
upstream OpenPose 1.7.0 ``src/openpose/pose/poseParameters.cpp`` (the version
named at ``openpose/native/tutorial_api_python/01_body_from_image.py:67``) and
are marked UNVERIFIED-vs-upstream-source: there is no OpenPose checkout in the
build container.  Consistency checks that do hold: ``TORSO_IDS`` of the
reference (``openpose/native/python/skeleton.py:11``) = Neck, RShoulder,
LShoulder, MidHip, RHip, LHip in this table, and the committed 25x3 result row
``data/test/output/inference_native/predictions.txt``.
"""

# --------------------------------------------------------------------------
# V1 (in-tree cv2 flavour) tables: openpose/opencv/common.py
# --------------------------------------------------------------------------

# openpose/opencv/common.py:2-21
COLORS = [
    [0, 100, 255], [0, 100, 255], [0, 255, 255], [0, 100, 255],
    [0, 255, 255], [0, 100, 255], [0, 255, 0], [255, 200, 100],
    [255, 0, 255], [0, 255, 0], [255, 200, 100], [255, 0, 255],
    [0, 0, 255], [255, 0, 0], [200, 200, 0], [255, 0, 0],
    [200, 200, 0], [0, 0, 0],
]

# openpose/opencv/common.py:26-28.  16 entries for 17 pose pairs: the
# LEye->LEar limb is never decoded by the reference.
MAPIDX = [[31, 32], [39, 40], [33, 34], [35, 36], [41, 42], [43, 44],
          [19, 20], [21, 22], [23, 24], [25, 26], [27, 28], [29, 30],
          [47, 48], [49, 50], [53, 54], [51, 52]]

_COCO_PARTS = ["Nose", "Neck", "RShoulder", "RElbow", "RWrist", "LShoulder",
               "LElbow", "LWrist", "RHip", "RKnee", "RAnkle", "LHip", "LKnee",
               "LAnkle", "REye", "LEye", "REar", "LEar", "Background"]
_COCO_PAIRS = [["Neck", "RShoulder"], ["Neck", "LShoulder"],
               ["RShoulder", "RElbow"], ["RElbow", "RWrist"],
               ["LShoulder", "LElbow"], ["LElbow", "LWrist"],
               ["Neck", "RHip"], ["RHip", "RKnee"], ["RKnee", "RAnkle"],
               ["Neck", "LHip"], ["LHip", "LKnee"], ["LKnee", "LAnkle"],
               ["Neck", "Nose"], ["Nose", "REye"], ["REye", "REar"],
               ["Nose", "LEye"], ["LEye", "LEar"]]
_MPI_PARTS = ["Head", "Neck", "RShoulder", "RElbow", "RWrist", "LShoulder",
              "LElbow", "LWrist", "RHip", "RKnee", "RAnkle", "LHip", "LKnee",
              "LAnkle", "Chest", "Background"]
_MPI_PAIRS = [["Head", "Neck"], ["Neck", "RShoulder"],
              ["RShoulder", "RElbow"], ["RElbow", "RWrist"],
              ["Neck", "LShoulder"], ["LShoulder", "LElbow"],
              ["LElbow", "LWrist"], ["Neck", "Chest"], ["Chest", "RHip"],
              ["RHip", "RKnee"], ["RKnee", "RAnkle"], ["Chest", "LHip"],
              ["LHip", "LKnee"], ["LKnee", "LAnkle"]]


def get_body_parts_and_pose_pairs(dataset=None):
    """Same contract as ``openpose/opencv/common.py:31-102``."""
    if dataset == 'COCO':
        body_parts = {n: i for i, n in enumerate(_COCO_PARTS)}
        pose_pairs = [list(p) for p in _COCO_PAIRS]
    elif dataset == 'MPI':
        body_parts = {n: i for i, n in enumerate(_MPI_PARTS)}
        pose_pairs = [list(p) for p in _MPI_PAIRS]
    else:
        raise(Exception("you need to specify either 'COCO', 'MPI' in dataset"))
    return body_parts, pose_pairs


# --------------------------------------------------------------------------
# V2 (upstream OpenPose semantics) tables.  UNVERIFIED vs upstream source.
# --------------------------------------------------------------------------

BODY_25_PARTS = ["Nose", "Neck", "RShoulder", "RElbow", "RWrist", "LShoulder",
                 "LElbow", "LWrist", "MidHip", "RHip", "RKnee", "RAnkle",
                 "LHip", "LKnee", "LAnkle", "REye", "LEye", "REar", "LEar",
                 "LBigToe", "LSmallToe", "LHeel", "RBigToe", "RSmallToe",
                 "RHeel", "Background"]

# limb (partA, partB) pairs, upstream POSE_BODY_25_PAIRS order
BODY_25_PAIRS = [(1, 8), (1, 2), (1, 5), (2, 3), (3, 4), (5, 6), (6, 7),
                 (8, 9), (9, 10), (10, 11), (8, 12), (12, 13), (13, 14),
                 (1, 0), (0, 15), (15, 17), (0, 16), (16, 18), (2, 17),
                 (5, 18), (14, 19), (19, 20), (14, 21), (11, 22), (22, 23),
                 (11, 24)]
# PAF channel pair per limb, offset from the first PAF channel (= 26)
BODY_25_MAPIDX = [(0, 1), (14, 15), (22, 23), (16, 17), (18, 19), (24, 25),
                  (26, 27), (6, 7), (2, 3), (4, 5), (8, 9), (10, 11),
                  (12, 13), (30, 31), (32, 33), (36, 37), (34, 35), (38, 39),
                  (20, 21), (28, 29), (40, 41), (42, 43), (44, 45), (46, 47),
                  (48, 49), (50, 51)]

# upstream COCO-18 (19 limbs; the two last are the ear-shoulder helpers)
COCO_18_PAIRS = [(1, 2), (1, 5), (2, 3), (3, 4), (5, 6), (6, 7), (1, 8),
                 (8, 9), (9, 10), (1, 11), (11, 12), (12, 13), (1, 0),
                 (0, 14), (14, 16), (0, 15), (15, 17), (2, 16), (5, 17)]
COCO_18_MAPIDX = [(12, 13), (20, 21), (14, 15), (16, 17), (22, 23), (24, 25),
                  (0, 1), (2, 3), (4, 5), (6, 7), (8, 9), (10, 11), (28, 29),
                  (30, 31), (34, 35), (32, 33), (36, 37), (18, 19), (26, 27)]


class PoseModelV2:
    """Constants of one upstream pose model (restated, see module docstring).

    nms_threshold / inter_* / min_subset_* follow upstream
    ``getPoseDefaultNmsThreshold`` etc. with ``maximizePositives=false``.
    """

    def __init__(self, name, parts, pairs, mapidx, ear_pair_indices, connector=1):
        self.name = name
        self.parts = parts
        self.n_parts = len(parts) - 1          # without background
        self.n_heat = len(parts)               # heatmap channels incl. bkg
        self.pairs = pairs
        self.mapidx = mapidx
        self.n_pairs = len(pairs)
        self.n_channels = self.n_heat + 2 * self.n_pairs
        self.ear_pair_indices = ear_pair_indices
        self.max_peaks = 127                   # POSE_MAX_PEOPLE
        self.nms_threshold = 0.05
        self.inter_min_above_threshold = 0.95
        self.inter_threshold = 0.05
        self.min_subset_cnt = 3
        self.min_subset_score = 0.4
        # person assembly of upstream bodyPartConnector: 1 = the CUDA build's pafPtrIntoVector +
        # pafVectorIntoPeopleVector (what the reference's Docker image runs, Dockerfile.U20:66-95), 0 = the
        # CPU-only build's per-limb greedy + createPeopleVector
        self.connector = connector

    def with_connector(self, connector):
        """The same model with the other upstream assembly route."""
        return PoseModelV2(self.name, self.parts, self.pairs, self.mapidx, self.ear_pair_indices, connector)


BODY_25 = PoseModelV2("BODY_25", BODY_25_PARTS, BODY_25_PAIRS, BODY_25_MAPIDX,
                      (18, 19))
COCO_18 = PoseModelV2("COCO", _COCO_PARTS, COCO_18_PAIRS, COCO_18_MAPIDX,
                      (17, 18))

BODY_25_CPU = BODY_25.with_connector(0)
COCO_18_CPU = COCO_18.with_connector(0)

V2_MODELS = {"BODY_25": BODY_25, "COCO": COCO_18}

# openpose/native/python/skeleton.py:11
TORSO_IDS = [1, 2, 5, 8, 9, 12]
