#!/bin/bash
# SQ counters of the decode tile kernel on the headline column (and the u32 tag column with "tags"),
# one rocprofv3 --pmc pass per counter group.  Usage: scripts/pmc_decode.sh [tags]
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/pmc_decode${1:+_$1}
mkdir -p "$out"
export TMPDIR=/tmp
cd /tmp
extra=${1:+--$1}
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD" \
           "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_LEVEL_WAVES"; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $grp --output-format csv -d "$out/p$i" -o p -- \
      python3 "$root/scripts/ab_decode.py" 2048:512:3:0:1 --rounds 1 --reps 3 $extra > /dev/null
done
python3 - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(list)
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "decode_tiles_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(out + "/summary.txt", "w") as w:
    for k in sorted(acc):
        v = acc[k]
        line = f"{k:28s} launches {len(v):3d} mean {sum(v)/len(v):16.1f}"
        print(line); w.write(line + "\n")
PY
