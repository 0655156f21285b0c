#!/bin/bash
# per-kernel durations (us) of one encode of a 2^27-row column; usage: trace_enc.sh label bits...
R=$GRAFT_REPO_ROOT; label=$1; shift
mkdir -p $R/gpurun_out/prof5; cd /tmp; export TMPDIR=/tmp
for b in "$@"; do
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof5 -o t$b -- python3 $R/scripts/enc_trace.py $b > $R/gpurun_out/prof5/run$b.log 2>&1
  python3 - <<PY
import csv
rows=[r for r in csv.DictReader(open("$R/gpurun_out/prof5/t${b}_kernel_trace.csv")) if "zbra" in r["Kernel_Name"]]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
rows=rows[len(rows)//2:]
print("$label bits $b", " ".join("%s=%.0f" % (r["Kernel_Name"][10:16], (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1000) for r in rows if int(r["End_Timestamp"])-int(r["Start_Timestamp"])>8000))
PY
done
