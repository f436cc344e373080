# mirrors openpose/native/python/test.py of the reference (test_op_runtime :9-48)
from posehip.bench_native import test_op_runtime  # noqa: F401

if __name__ == "__main__":
    test_op_runtime()
    print("[INFO] : FINISHED")
