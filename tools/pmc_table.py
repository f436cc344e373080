#!/usr/bin/env python3
"""Per-kernel table from a rocprofv3 `--pmc ... --output-format csv` run (one pass of SQ counters).

    python tools/pmc_table.py <dir with *_counter_collection.csv> [name filter regex] > table.md

Counters expected (any subset): SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES.  Values are means per launch,
summed over XCDs, in millions.  Derived (as in profiles/r01_pmc_decode_counters.md):
  clock          = (SQ_BUSY_CYCLES / 32) / duration          (SQ_BUSY_CYCLES counts once per shader engine, 32 of them)
  VALU pipe use  = SQ_ACTIVE_INST_VALU x 4 / (SQ_BUSY_CYCLES / 32 x 1024 SIMDs)
  MFMA busy      = SQ_VALU_MFMA_BUSY_CYCLES / (SQ_BUSY_CYCLES / 32 x 1024 SIMDs)
  LDS use        = SQ_LDS_IDX_ACTIVE / (SQ_BUSY_CYCLES / 32 x 256 CUs)
  wait / wave    = SQ_WAIT_ANY / SQ_WAVE_CYCLES
"""
import csv
import glob
import re
import sys
from collections import defaultdict


def main():
    d = sys.argv[1]
    flt = re.compile(sys.argv[2]) if len(sys.argv) > 2 else None
    files = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    acc = defaultdict(lambda: defaultdict(float))   # kernel -> counter -> sum
    disp = defaultdict(dict)                         # kernel -> dispatch id -> duration ns
    shape = {}
    for f in files:
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if flt and not flt.search(k):
                continue
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            disp[k][r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
            shape[k] = f"{int(r['Grid_Size']) // max(1, int(r['Workgroup_Size']))} x {r['Workgroup_Size']}"
    print("| kernel | launches | duration | BUSY_CYCLES/32 | clock | VALU pipe use | MFMA busy | LDS use | LDS conflict / active "
          "| wait / wave | grid x block |")
    print("|---|---|---|---|---|---|---|---|---|---|---|")
    rows = []
    for k, c in acc.items():
        n = len(disp[k])
        dur = sum(disp[k].values()) / n            # ns
        g = lambda name: c.get(name, 0.0) / n      # noqa: E731
        busy = g("SQ_BUSY_CYCLES") / 32.0
        pct = lambda v: f"{100 * v:.0f} %" if busy > 0 and v == v else "-"   # noqa: E731
        valu = g("SQ_ACTIVE_INST_VALU") * 4 / (busy * 1024) if busy and "SQ_ACTIVE_INST_VALU" in c else float("nan")
        mfma = g("SQ_VALU_MFMA_BUSY_CYCLES") / (busy * 1024) if busy and "SQ_VALU_MFMA_BUSY_CYCLES" in c else float("nan")
        lds = g("SQ_LDS_IDX_ACTIVE") / (busy * 256) if busy and "SQ_LDS_IDX_ACTIVE" in c else float("nan")
        conf = g("SQ_LDS_BANK_CONFLICT") / g("SQ_LDS_IDX_ACTIVE") if g("SQ_LDS_IDX_ACTIVE") else float("nan")
        wait = g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES") if g("SQ_WAVE_CYCLES") else float("nan")
        rows.append((dur * n, f"| `{k[:70]}` | {n} | {dur / 1e3:.1f} us | {busy / 1e6:.3f} M | "
                              f"{busy / dur:.2f} GHz | {pct(valu)} | {pct(mfma)} | {pct(lds)} | "
                              f"{conf:.2f} | {wait:.2f} | {shape[k]} |"))
    for _, line in sorted(rows, reverse=True):
        print(line)


if __name__ == "__main__":
    main()
