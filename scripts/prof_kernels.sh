#!/bin/bash
# Per-kernel average times of a python script under rocprofv3.  Usage: prof_kernels.sh script.py args...
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
d=$(mktemp -d /tmp/prof.XXXXXX)
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$d" -o t -- python3 "$root/$1" "${@:2}" 2>/dev/null | tail -1
python3 - "$d" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "zbra" in r["Name"]:
        print("  %-52s %4s x %9.1f us  (min %9.1f, max %9.1f)" % (r["Name"].split("(")[0][:52], r["Calls"], float(r["AverageNs"]) / 1000,
                                                                   float(r["MinNs"]) / 1000, float(r["MaxNs"]) / 1000))
PY
rm -rf "$d"
