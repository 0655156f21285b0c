#!/bin/bash
# Encode of u32 tag arrays with the u32 pack kernel (4096 tags per workgroup; default) against the
# generic one (ZBRA_HIP_PACK_U32=0): the 128 tag arrays of 2^22 rows of the sum-typed table as one
# batch, and one 2^29-row tag column.  Alternating processes on one box.
set -e
cd "$(dirname "$0")/.."
for round in 1 2 3; do
    for on in 1 0; do
        echo "ZBRA_HIP_PACK_U32=$on"
        ZBRA_HIP_PACK_U32=$on timeout -k 10 200 python3 scripts/enc_tags_rate.py
    done
done
