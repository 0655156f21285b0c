#!/bin/bash
# Per-kernel times and HBM traffic of one encode width.  Usage: prof_encode_width.sh <amp_bits> <tag>
# Writes gpurun_out/<tag>_encode_w<bits>.txt: rocprofv3 --kernel-trace --stats per-kernel averages,
# then FETCH_SIZE and WRITE_SIZE per kernel from two --pmc runs of their own (FETCH_SIZE doubled:
# gfx950 tallies the 128-B requests of 16-B-per-lane loads at 64 B).
set -e
bits=$1; tag=${2:-r02}
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/${tag}_encode_w${bits}.txt
d=$(mktemp -d /tmp/prof.XXXXXX)
export TMPDIR=/tmp
cd /tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$d/t" -o t -- python3 "$root/scripts/enc_width_prof.py" "$bits" ${ROWS_LOG2:-27} 10 2>/dev/null | tail -1 > "$out"
if [ -z "$FAST" ]; then  # FAST=1: kernel times only
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$d/f" -o f -- python3 "$root/scripts/enc_width_prof.py" "$bits" ${ROWS_LOG2:-27} 2 > /dev/null 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$d/w" -o w -- python3 "$root/scripts/enc_width_prof.py" "$bits" ${ROWS_LOG2:-27} 2 > /dev/null 2>&1
fi
python3 - "$d" "$bits" "$out" >> "$out" <<'PY'
import csv, glob, re, sys
from collections import defaultdict
d, bits = sys.argv[1], int(sys.argv[2])
n = 1 << 27
short = lambda s: re.sub(r"\(.*", "", s).replace("void ", "")
times = {}
for r in csv.DictReader(open(glob.glob(d + "/t/**/*kernel_stats.csv", recursive=True)[0])):
    if "zbra" in r["Name"]:
        times[short(r["Name"])] = (int(r["Calls"]), float(r["AverageNs"]) / 1000)
def pmc(sub):
    per = defaultdict(list)
    if not glob.glob(d + f"/{sub}/**/*counter_collection.csv", recursive=True):
        return per
    for r in csv.DictReader(open(glob.glob(d + f"/{sub}/**/*counter_collection.csv", recursive=True)[0])):
        if "zbra" in r["Kernel_Name"]:
            per[short(r["Kernel_Name"])].append(float(r["Counter_Value"]) * 1024)   # KiB -> B
    return per
f, w = pmc("f"), pmc("w")
print("%-34s %6s %10s %14s %14s" % ("kernel", "calls", "avg us", "read MB/launch", "written MB/launch"))
tot_t = tot_r = tot_w = 0.0
per_encode = 20  # launches of the script: 2 x reps(10)
for k, (calls, us) in sorted(times.items(), key=lambda kv: -kv[1][0] * kv[1][1]):
    rd = 2 * sum(f.get(k, [0])) / max(1, len(f.get(k, [0])))
    wr = sum(w.get(k, [0])) / max(1, len(w.get(k, [0])))
    mult = calls / per_encode
    tot_t += us * mult; tot_r += rd * mult; tot_w += wr * mult
    print("%-34s %6d %10.1f %14.1f %14.1f" % (k[:34], calls, us, rd / 1e6, wr / 1e6))
print("per encode: kernels %.1f us, HBM read %.1f MB + written %.1f MB = %.1f MB" % (tot_t, tot_r / 1e6, tot_w / 1e6, (tot_r + tot_w) / 1e6))
algo = re.search(r"algorithmic (\d+) B", open(sys.argv[3]).readline())
if algo and tot_r + tot_w > 0:
    print("HBM traffic / algorithmic bytes (two reads of the values + the encoding) = %.3f" % ((tot_r + tot_w) / int(algo.group(1))))
PY
rm -rf "$d"
cat "$out"
