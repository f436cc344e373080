# openpose/native/python/inference.py: the pipeline callers; `python -m openpose.native.python.inference --op-mode ...`
from posehip.inference import (ExtractSkeletonAndTrack, compare_tracker, extract_2dskeletons,  # noqa: F401
                               extract_2dskeletons_and_track_offline, extract_2dskeletons_offline,
                               extract_2dskeletons_offline_batched, extract_2dskeletons_online, main)

if __name__ == "__main__":
    main()
