#!/bin/bash
# Counters of the kernels whose name contains a given string, one rocprofv3 --pmc pass per counter
# group (three SQ groups, FETCH_SIZE, WRITE_SIZE; FETCH_SIZE is shown doubled: gfx950 tallies the
# 128-byte requests of 16-byte-per-lane loads at 64 bytes).
# Usage: scripts/pmc_kernel.sh <name-substring[,another]> <tag> script.py [args...]
# Writes gpurun_out/<tag>_pmc.txt.
set -e
names=$1; tag=$2; shift 2
root=$(cd "$(dirname "$0")/.." && pwd)
d=$(mktemp -d /tmp/pmc.XXXXXX)
export TMPDIR=/tmp
cd /tmp
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD" \
           "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -s KILL 150 rocprofv3 --pmc $grp --output-format csv -d "$d/p$i" -o p -- python3 "$root/scripts/$1" "${@:2}" > /dev/null 2>&1
done
python3 - "$d" "$names" > "$root/gpurun_out/${tag}_pmc.txt" <<'PY'
import csv, glob, sys, collections, re
d, names = sys.argv[1], sys.argv[2].split(",")
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(d + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
        if any(n in k for n in names):
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    c = {n: sum(v) / len(v) for n, v in acc[k].items()}
    print(k)
    waves = c.get("SQ_WAVES", 0) or 1
    for n in sorted(c):
        extra = ""
        if n == "FETCH_SIZE": extra = f"   = {2 * c[n] * 1024 / 1e6:10.1f} MB read per launch (doubled)"
        elif n == "WRITE_SIZE": extra = f"   = {c[n] * 1024 / 1e6:10.1f} MB written per launch"
        elif n.startswith(("SQ_INSTS", "SQ_WAVE_CYCLES", "SQ_WAIT", "SQ_ACTIVE")): extra = f"   {c[n] / waves:10.1f} per wave"
        print(f"  {n:26s} {c[n]:18.1f}{extra}")
PY
rm -rf "$d"
cat "$root/gpurun_out/${tag}_pmc.txt"
