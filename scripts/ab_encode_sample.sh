#!/bin/bash
# Encoder builds with a larger median sample (build_ab/libzbra_s8k.so, libzbra_s16k.so: see the
# ZBRA_SAMPLE block of zbra_encode.hpp) against the default, alternating, one process per run.
set -e
cd "$(dirname "$0")/.."
for round in 1 2; do
    for lib in zbra_amd/libzbra_hip.so build_ab/libzbra_s8k.so build_ab/libzbra_s16k.so; do
        ZBRA_HIP_LIB=$PWD/$lib timeout -k 10 150 python3 scripts/encode_sweep.py "$(basename $lib .so | sed s/libzbra_//)"
    done
done
