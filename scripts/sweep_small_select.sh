#!/bin/bash
# Encode times of the batch configurations against the longest select source one workgroup
# finishes on its own (ZBRA_HIP_SMALL_SELECT_KEYS).  Run from the repository root on the GPU box.
for k in 4096 16384 32768 65536 131072 524288; do
    export ZBRA_HIP_SMALL_SELECT_KEYS=$k
    echo "== small select keys $k"
    timeout -k 10 200 python3 scripts/bench_configs.py --config 3 4 --reps 5 | grep -o '"case[^,]*\|encode_ms[^,]*' || exit 1
    timeout -k 10 100 python3 scripts/enc_batch_rate.py 96 22 w40 || exit 1
    timeout -k 10 100 python3 scripts/enc_batch_rate.py 96 22 time || exit 1
    timeout -k 10 100 python3 scripts/enc_batch_rate.py 16 24 w40 || exit 1
done
