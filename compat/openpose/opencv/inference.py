# mirrors openpose/opencv/inference.py of the reference (writers :17-30, timing loop :33-89)
from posehip.bench_opencv import run, save_keypoints, save_personwise_keypoints  # noqa: F401

if __name__ == "__main__":
    run()
