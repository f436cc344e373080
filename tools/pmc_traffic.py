#!/usr/bin/env python3
"""Builds profiles/rNN_hbm_traffic.json from two rocprofv3 PMC passes (run separately, as the MI355X guide
prescribes) and one --kernel-trace --stats pass of the same command:

    rocprofv3 --pmc FETCH_SIZE -d gpurun_out/pmc_fetch -- python3 bench.py --streams 1 --steps 3 --warmup 1 --cpu-frames 0
    rocprofv3 --pmc WRITE_SIZE -d gpurun_out/pmc_write -- python3 bench.py --streams 1 --steps 3 --warmup 1 --cpu-frames 0
    rocprofv3 --kernel-trace --stats -d gpurun_out/stats -- python3 bench.py --streams 1 --steps 5 --warmup 2 --cpu-frames 0
    python tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/stats profiles/r01_hbm_traffic.json

FETCH_SIZE / WRITE_SIZE are in KB; FETCH_SIZE is doubled (gfx950 tallies 128-B requests at 64 B).
"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"^void ", "", name)
    return re.sub(r"\(.*$", "", name)


def counter_mean(folder, counter):
    tot, cnt = defaultdict(float), defaultdict(int)
    for path in glob.glob(os.path.join(folder, "**", "*counter_collection.csv"), recursive=True):
        with open(path) as f:
            for row in csv.DictReader(f):
                if row["Counter_Name"] == counter:
                    k = short(row["Kernel_Name"])
                    tot[k] += float(row["Counter_Value"])
                    cnt[k] += 1
    return {k: tot[k] / cnt[k] for k in tot}


def stats(folder):
    out = {}
    for path in glob.glob(os.path.join(folder, "**", "*kernel_stats.csv"), recursive=True):
        with open(path) as f:
            for row in csv.DictReader(f):
                out[short(row["Name"])] = float(row["AverageNs"])
    return out


def main():
    fetch_dir, write_dir, stats_dir, out_path = sys.argv[1:5]
    fetch = counter_mean(fetch_dir, "FETCH_SIZE")
    write = counter_mean(write_dir, "WRITE_SIZE")
    avg = stats(stats_dir)
    conv = {k: int(round(fetch.get(k, 0.0) * 2 * 1024 + write.get(k, 0.0) * 1024))
            for k in sorted(fetch) if k.startswith("conv")}
    tail = {}
    for k in sorted(fetch):
        if k.startswith(("conv", "__amd", "maxpool", "ph_copy")) or k not in avg:
            continue
        fb, wb = fetch[k] * 2 * 1024, write.get(k, 0.0) * 1024
        tail[k] = {"fetch_bytes_x2": int(round(fb)), "write_bytes": int(round(wb)), "avg_ns": int(round(avg[k])),
                   "achieved_GB_per_s": round((fb + wb) / avg[k], 2)}
    doc = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 bench.py --streams 1 "
                     "--steps 3 --warmup 1 --cpu-frames 0; FETCH_SIZE (KB) doubled per MI355X_MICROARCH.md (gfx950 "
                     "tallies 128-B requests at 64 B), WRITE_SIZE (KB) as read; mean over the launches of each "
                     "instantiation; tools/pmc_traffic.py",
           "bytes_per_launch": conv, "tail_kernels": tail}
    with open(out_path, "w") as f:
        json.dump(doc, f, indent=1)
    print(json.dumps(conv, indent=1))


if __name__ == "__main__":
    main()
