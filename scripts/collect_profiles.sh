#!/bin/bash
# Kernel-level profiles beyond what profile_bench.sh collects: per-kernel times of the other configurations and of the widened rows, and the encode of a
# wide (25-bit) and a full-range (64-bit) column with HBM bytes per kernel.
# Usage: scripts/collect_profiles.sh r03   (writes gpurun_out/<tag>_*.txt; copy what is to be kept into profiles/)
tag=${1:-r03}
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out
cd "$root"
bash scripts/prof_encode_width.sh 24 $tag > /dev/null 2>&1
bash scripts/prof_encode_width.sh 64 $tag > /dev/null 2>&1
bash scripts/prof_kernels.sh scripts/offsets_rate.py > $out/${tag}_offsets_kernels.txt 2>&1
bash scripts/prof_kernels.sh scripts/enum_rate.py > $out/${tag}_enum_kernels.txt 2>&1
bash scripts/prof_kernels.sh scripts/bench_configs.py --config 3 --reps 5 > $out/${tag}_config3_kernels.txt 2>&1
bash scripts/prof_kernels.sh scripts/bench_configs.py --config 4 --reps 5 > $out/${tag}_config4_kernels.txt 2>&1
bash scripts/prof_kernels.sh scripts/bench_file.py > $out/${tag}_file_kernels.txt 2>&1
ls -la $out/${tag}_*
