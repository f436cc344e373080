#!/usr/bin/env python3
"""Copies what tools/refresh_profiles.sh left under gpurun_out/prof into the tracked profiles/r03_* files:
whole files where a file is one run's output, sections (between two '# ' headings, or a counter table) where a file
is a collection with hand-written readings around the tables.  python3 tools/collect_profiles.py [gpurun_out/prof]"""
import glob, os, re, shutil, sys
src = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/prof"
dst = "profiles"


def read(p):
    return open(os.path.join(src, p)).read()


def json_line(text):   # the bench prints one JSON line; stderr notes may surround it
    return [l for l in text.splitlines() if l.startswith("{")][-1] + "\n"


open(f"{dst}/r03_bench.json", "w").write(json_line(read("bench.json")))
open(f"{dst}/r03_bench_1080p.json", "w").write(json_line(read("bench_1080p.json")))
shutil.copy(f"{src}/coco_mot17.json", f"{dst}/r03_coco_mot17.json")
shutil.copy(f"{src}/hbm_traffic.json", f"{dst}/r03_hbm_traffic.json")
for sub, name in (("stats", "r03_bench_kernel_stats.csv"), ("stats3", "r03_bench_kernel_stats_3inflight.csv"),
                  ("v1stats", "r03_coco_mot17_kernel_stats.csv")):
    f = glob.glob(f"{src}/{sub}/*/*kernel_stats.csv")
    if f:
        shutil.copy(f[0], f"{dst}/{name}")

# r03_tools.txt: replace the body of the sections these tools own
tools = open(f"{dst}/r03_tools.txt").read()
own = {"# tools/decode_bench.py": "decode_bench.txt", "# tools/overlap_probe.py": "overlap_probe.txt",
       "# tools/conv_bench.py": "conv_bench.txt", "# tools/stem_ab.py": "stem_ab.txt",
       "# tools/latency_bench.py": "latency_bench.txt", "# tools/batch_sweep.py": "batch_sweep.txt"}
parts = re.split(r"(?m)^(?=# )", tools)
out = []
for part in parts:
    head = part.split("\n", 1)[0]
    hit = [k for k in own if head.startswith(k)]
    if hit:
        body = "\n".join(l for l in read(own[hit[0]]).splitlines() if l.strip() and "amdgpu.ids" not in l)
        part = head + "\n" + body + "\n\n"
    elif head.startswith("# tools/coco_mot17_bench.py"):
        body = "\n".join(l for l in read("coco_mot17.log").splitlines() if l.startswith("{"))
        part = head + "\n" + body + "\n\n"
    out.append(part)
open(f"{dst}/r03_tools.txt", "w").write("".join(out))


def swap_table(path, marker, new_table):
    """replace the first markdown table after `marker` in `path`"""
    s = open(path).read()
    i = s.index(marker)
    a = s.index("\n|", i) + 1
    m = re.compile(r"(?m)^(?!\|)").search(s, a)
    b = m.start()
    open(path, "w").write(s[:a] + new_table.strip() + "\n" + s[b:])


def strip_args(table):   # `kernel(args...` -> `kernel`
    return re.sub(r"`([A-Za-z0-9_<>, ]+?)\([^`]*`", r"`\1`", table)


swap_table(f"{dst}/r03_pmc_decode_counters.md", "## V2 decode", strip_args(read("pmc_v2.md")))
swap_table(f"{dst}/r03_pmc_decode_counters.md", "## V1 decode", strip_args(read("pmc_v1.md")))
print("collected; r03_pmc_conv_counters.md is edited by hand from", f"{src}/pmc_conv.md")
