#!/bin/bash
# Prints the bench line's encode and decode timings. Usage: bash scripts/enc_times.sh [label]
out=$(timeout -k 10 100 python -u bench.py --steps 30 --warmup 5 --cpu-seconds 0 2>/dev/null) || { echo "$1 FAILED"; exit 1; }
echo "$out" | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$1', 'encode', d['encode'], 'decode_ms', d['roofline']['kernel_ms'], d['roofline']['frac'])"
