#!/bin/bash
# Batched encodes (the mixed-width column cut into 2^20-row arrays; the sum-typed table's columns)
# against the grid of the select's level kernels and the tiles a workgroup takes of a sparse level.
root=$(cd "$(dirname "$0")/.." && pwd)
cd "$root"
for g in ${GRIDS:-1024 2048 4096 8192}; do
  for t in ${TILES:-2 4 8 16 32}; do
    echo "grid $g tiles $t: $(ZBRA_HIP_LEVEL_GRID=$g ZBRA_HIP_SPARSE_TILES=$t timeout -k 10 200 python3 scripts/bench_configs.py --encode-only --reps 20 2>/dev/null | python3 -c "
import json, sys
print(' | '.join('%s %.3f' % (json.loads(l)['case'][:12], json.loads(l)['encode_ms']) for l in sys.stdin if l.startswith('{')))")" || exit 1
  done
done
