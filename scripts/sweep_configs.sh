#!/bin/bash
# bench_configs.py (config 4 tags / ints / both, config 3) per decode variant. Usage: sweep_configs.sh "v1 v2" [xcd]
for v in $1; do
  ZBRA_HIP_DECODE_VARIANT=$v ZBRA_HIP_DECODE_XCD=${2:-1} timeout -k 10 300 python -u scripts/bench_configs.py --config 3 4 --reps 6 2>/dev/null | python -c "
import sys,json
print('$v xcd=${2:-1}', ' | '.join('%.3f ms %.3f' % (json.loads(l)['decode_ms'], json.loads(l)['hbm_frac']) for l in sys.stdin))" || exit 1
done
