#!/bin/bash
# Encode of u32 tag arrays: u32 forms of the bracket pass and of the pack kernel (defaults) against the
# generic kernels (ZBRA_HIP_BRACKET_U32=0 / ZBRA_HIP_PACK_U32=0).  Alternating processes on one box.
set -e
cd "$(dirname "$0")/.."
for round in 1 2 3; do
    for cfg in "1 1" "0 1" "0 0"; do
        set -- $cfg
        echo "bracket_u32=$1 pack_u32=$2"
        ZBRA_HIP_BRACKET_U32=$1 ZBRA_HIP_PACK_U32=$2 timeout -k 10 200 python3 scripts/enc_tags_rate.py
    done
done
