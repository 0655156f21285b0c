#!/bin/bash
# Runs bench.py once per decode-kernel variant (tile:nt:direct) and prints kernel ms / GB/s.
# Usage on the GPU box: bash scripts/sweep_decode.sh "512:0:0 512:1:0 ..."
set -o pipefail
for v in $1; do
  out=$(ZBRA_HIP_DECODE_VARIANT=$v timeout -k 10 100 python -u bench.py --steps 40 --warmup 5 --cpu-seconds 0 --no-encode-bench 2>/dev/null) || { echo "$v FAILED"; exit 1; }
  echo "$out" | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$v', d['roofline']['kernel_ms'], d['roofline']['achieved'], d['value'])"
done
