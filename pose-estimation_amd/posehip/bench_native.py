"""The native-flavour timing harness through ``PyOpenPoseNative``
(counterpart of ``test_op_runtime``, ``openpose/native/python/test.py:9-48``).

    python -m posehip.bench_native [--image ...] [--iters 500] [--random-weights 0]

500 x (``predict`` + ``filter_prediction``) on one 384x384 frame; the first half of the iterations warms up, the
second half is timed (the reference's ``n > N // 2``), and seconds per frame and frames per second are printed.
"""
import argparse
import os
import time

from .bench_opencv import load_frame


def test_op_runtime(image_path="data/test/pexels-photo-4384679.jpeg", target_path="data/test/output/inference_native",
                    model_folder="/usr/local/src/openpose/models/", model_pose="BODY_25", net_resolution="-1x368",
                    skel_thres=0.5, max_true_body=2, patch_offset=2, iters=500, random_weights=None, save=False,
                    quiet=False):
    """-> (seconds per frame, frames per second, the estimator)."""
    from .api_native import PyOpenPoseNative
    os.makedirs(target_path, exist_ok=True)
    params = dict(model_folder=model_folder, model_pose=model_pose, net_resolution=net_resolution)
    if random_weights is not None:
        params["posehip_random_weights"] = random_weights
    pyop = PyOpenPoseNative(params, skel_thres, max_true_body, patch_offset)
    image = load_frame(image_path, 384)
    timed, spent = 0, 0.0
    for n in range(iters):
        t_start = time.time()
        pyop.predict(image)
        pyop.filter_prediction()
        if n > iters // 2:
            spent += time.time() - t_start
            timed += 1
    if save:
        pyop.save_pose_keypoints(f'{target_path}/predictions.txt')
    per_frame = spent / max(timed, 1)
    fps = timed / spent if spent > 0 else float("inf")
    if not quiet:
        print(f"\nAverage inference time over {iters - iters // 2} trials : {per_frame:.6f}s or {fps:.6f}fps")
        print(pyop.pose_scores)
    return per_frame, fps, pyop


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--image", default="data/test/pexels-photo-4384679.jpeg")
    ap.add_argument("--iters", type=int, default=500)
    ap.add_argument("--random-weights", type=int, default=None)
    ap.add_argument("--save", action="store_true", help="write predictions.txt as the reference's commented line does")
    a = ap.parse_args()
    test_op_runtime(a.image, iters=a.iters, random_weights=a.random_weights, save=a.save)
    print("[INFO] : FINISHED")
