"""The cv2-flavour timing harness and its text writers, through the estimator class
(counterpart of ``openpose/opencv/inference.py``: ``save_keypoints`` :17-22, ``save_personwise_keypoints`` :25-30,
the 100-iteration loop :33-89).

    python -m posehip.bench_opencv [--image data/test/pillar_text.jpg] [--iters 100] [--random-weights 0]

As in the reference the clock starts after the image has been read and resized to 480x480 and covers ``predict`` +
``postprocess_multi`` + the append of the keypoint line; the printed figure is the mean over the iterations.
"""
import argparse
import os
import time

import numpy as np


def _append_line(save_path: str, values) -> None:
    """One comma-separated line of Python-float reprs, appended (the file is created when missing)."""
    with open(save_path, 'a+') as f:
        f.write(",".join(map(str, values)) + "\n")


def save_keypoints(keypoints_list: np.ndarray, save_path: str):
    """``keypoints_list`` f64 [K, 3] -> "x0,y0,s0,x1,y1,s1,..." as one appended line."""
    _append_line(save_path, np.asarray(keypoints_list).reshape(-1).tolist())


def save_personwise_keypoints(personwise_keypoints: list, save_path: str):
    """list of f64 rows (keypoint ids, -1 = missing, then the person's score) -> all rows on one appended line."""
    _append_line(save_path, [v for row in personwise_keypoints for v in np.asarray(row).tolist()])


def load_frame(image_path: str, side: int = 480) -> np.ndarray:
    """BGR u8 [side, side, 3]: the file decoded with PIL (as utils_rs.read_color_file does) and resized; a seeded
    noise frame when the file does not exist (the reference's own test image is not in its tree)."""
    if image_path and os.path.exists(image_path):
        from PIL import Image
        rgb = Image.open(image_path).convert("RGB").resize((side, side), Image.BILINEAR)
        return np.ascontiguousarray(np.asarray(rgb)[:, :, ::-1])
    return np.random.default_rng(0).integers(0, 256, (side, side, 3), dtype=np.uint8)


def run(image_path="data/test/pillar_text.jpg", target_path="openpose/output/inference_opencv", iters=100,
        proto_file="/data/openpose/opencv_data/models/pose/coco/pose_deploy_linevec.prototxt",
        weights_file="/data/openpose/opencv_data/models/pose/coco/pose_iter_440000.caffemodel", dataset="COCO",
        image_width=368, image_height=368, random_weights=None, quiet=False):
    """-> (mean seconds per iteration, last keypoints_list, last personwise_keypoints)."""
    from .api_opencv import PyOpenPoseOpenCV, get_body_parts_and_pose_pairs
    os.makedirs(target_path, exist_ok=True)
    body_parts, pose_pairs = get_body_parts_and_pose_pairs(dataset)
    pyop = PyOpenPoseOpenCV(image_height, image_width, body_parts, pose_pairs)   # (height, width), as the reference
    pyop.load(proto_file, weights_file, False, random_weights=random_weights)
    spent = 0.0
    keypoints_list, personwise_keypoints = np.zeros((0, 3)), []
    for _ in range(iters):
        image = load_frame(image_path)                                            # outside the timed region
        t_start = time.time()
        pred = pyop.predict(image)
        keypoints_list, personwise_keypoints = pyop.postprocess_multi(pred)
        save_keypoints(keypoints_list, f'{target_path}/kptl{int(time.time() * 1e8)}.txt')
        spent += time.time() - t_start
    mean = spent / max(iters, 1)
    if not quiet:
        print(f"Average inference time over {iters} trials : {mean}s")
    return mean, keypoints_list, personwise_keypoints


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--image", default="data/test/pillar_text.jpg")
    ap.add_argument("--out", default="openpose/output/inference_opencv")
    ap.add_argument("--iters", type=int, default=100)
    ap.add_argument("--random-weights", type=int, default=None,
                    help="seed: run the COCO graph with random weights when no model files exist")
    a = ap.parse_args()
    run(a.image, a.out, a.iters, random_weights=a.random_weights)
