#!/usr/bin/env python3
"""Turns the two rocprofv3 counter CSVs (FETCH_SIZE, WRITE_SIZE; KiB per dispatch) into
<tag>_pmc_traffic.json: HBM bytes per launch of the decode tile kernel, beside the algorithmic
bytes.  Usage: scripts/pmc_summary.py <dir> <tag>"""
import csv
import datetime
import json
import re
import sys
from collections import defaultdict

d, tag = sys.argv[1], sys.argv[2]
N = 1 << 28
ALGO_READ, ALGO_WRITE = 13 + N * 12 // 8, 8 * N


def short(name):
    return re.sub(r"\(.*", "", name)


def load(kind):
    per = defaultdict(list)
    with open(f"{d}/{tag}_pmc_{kind}_counter_collection.csv") as f:
        for row in csv.DictReader(f):
            if not row["Kernel_Name"].startswith(("zbra::", "void zbra::")):
                continue
            per[short(row["Kernel_Name"])].append(float(row["Counter_Value"]))
    return per


fetch, write = load("fetch"), load("write")
dec = next(k for k in fetch if "decode_tiles_kernel" in k)
raw_f = sum(fetch[dec]) / len(fetch[dec]) * 1024
raw_w = sum(write[dec]) / len(write[dec]) * 1024
read, written = 2 * raw_f, raw_w


def table(per):
    return {k: {"launches": len(v), "mean_KiB": sum(v) / len(v), "min_KiB": min(v), "max_KiB": max(v)} for k, v in per.items()}


json.dump({
    "collected": datetime.datetime.now(datetime.timezone.utc).strftime("%Y-%m-%d"),
    "command": "rocprofv3 --pmc FETCH_SIZE (then, in a run of its own, --pmc WRITE_SIZE) --output-format csv -- "
               "python3 bench.py --steps 5 --warmup 2 --cpu-seconds 0",
    "kernel": dec,
    "raw_counter_bytes": {"FETCH_SIZE": raw_f, "WRITE_SIZE": raw_w},
    "gfx950_correction": "FETCH_SIZE tallies 128-B read requests at 64 B for 16-B-per-lane streaming loads: doubled. "
                         "WRITE_SIZE exact for 16-B-per-lane stores.",
    "hbm_bytes_per_launch": {"read": read, "written": written, "total": read + written},
    "algorithmic_bytes_per_launch": {"read": ALGO_READ, "written": ALGO_WRITE, "total": ALGO_READ + ALGO_WRITE},
    "traffic_over_algorithmic": (read + written) / (ALGO_READ + ALGO_WRITE),
    "per_kernel": {"FETCH_SIZE": table(fetch), "WRITE_SIZE": table(write)},
}, open(f"{d}/{tag}_pmc_traffic.json", "w"), indent=1)
print(open(f"{d}/{tag}_pmc_traffic.json").read()[:900])
