#!/bin/bash
# The last <count> kernel dispatches of a python script, in launch order: duration and the gap to
# the dispatch before, in us.  Usage: trace_tail.sh <count> script.py args...
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
count=$1; shift
d=$(mktemp -d /tmp/prof.XXXXXX)
export TMPDIR=/tmp
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d "$d" -o t -- python3 "$root/$1" "${@:2}" 2>/dev/null | tail -1
python3 - "$d" "$count" <<'PY'
import csv, glob, re, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-int(sys.argv[2]):]
prev = None
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")[:44]
    print("  %-44s %9.1f us   gap %7.1f   grid %s" % (name, (e - s) / 1000, (s - prev) / 1000 if prev else 0.0, r.get("Grid_Size_X", r.get("Grid_Size", "?"))))
    prev = e
PY
rm -rf "$d"
