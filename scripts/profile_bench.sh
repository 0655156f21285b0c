#!/bin/bash
# Collects what profiles/ holds for one round, on the GPU box:
#   1. rocprofv3 --kernel-trace --stats of bench.py   -> <tag>_bench_kernel_stats.csv, <tag>_bench_under_rocprof.json
#   2. --pmc FETCH_SIZE and --pmc WRITE_SIZE, one run each -> <tag>_pmc_{fetch,write}_counter_collection.csv
#   3. a plain bench.py run                              -> <tag>_bench_line.json
# Usage: scripts/profile_bench.sh r01   (writes under gpurun_out/profiles_<tag>/; every step has its own time limit)
set -e
tag=${1:-r01}
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/profiles_$tag
mkdir -p "$out"
export TMPDIR=/tmp
cd /tmp
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -o trace -- \
    python3 "$root/bench.py" --steps 30 --warmup 5 --cpu-seconds 0 > "$out/${tag}_bench_under_rocprof.json"
cp "$(find "$out/trace" -name '*kernel_stats.csv' | head -n 1)" "$out/${tag}_bench_kernel_stats.csv"
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$out/fetch" -o fetch -- \
    python3 "$root/bench.py" --steps 5 --warmup 2 --cpu-seconds 0 > "$out/pmc_fetch_stdout.json"
cp "$(find "$out/fetch" -name '*counter_collection.csv' | head -n 1)" "$out/${tag}_pmc_fetch_counter_collection.csv"
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$out/write" -o write -- \
    python3 "$root/bench.py" --steps 5 --warmup 2 --cpu-seconds 0 > "$out/pmc_write_stdout.json"
cp "$(find "$out/write" -name '*counter_collection.csv' | head -n 1)" "$out/${tag}_pmc_write_counter_collection.csv"
rm -rf "$out/trace" "$out/fetch" "$out/write"
cd "$root"
timeout -k 10 300 python3 bench.py > "$out/${tag}_bench_line.json"
cat "$out/${tag}_bench_line.json"
