#!/bin/bash
# Encode of a single array with the finalize step done by the pack kernel's workgroups (default)
# against the separate finalize launch (ZBRA_HIP_ENCODE_FOLD=0): alternating processes on one box.
set -e
cd "$(dirname "$0")/.."
for round in 1 2 3; do
    for fold in 1 0; do
        ZBRA_HIP_ENCODE_FOLD=$fold timeout -k 10 150 python3 scripts/encode_sweep.py "fold=$fold"
    done
done
