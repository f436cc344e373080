# openpose/native/python/args.py: get_parser (same flags and defaults, plus --posehip-*)
from posehip.args import get_parser  # noqa: F401
