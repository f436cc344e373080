from posehip.tables import COLORS, MAPIDX, get_body_parts_and_pose_pairs  # noqa: F401
