#!/bin/bash
# Times the encode of a few columns and of configurations 3 and 4 with each library build given
# (paths to libzbra_hip.so variants; "default" = the in-tree build).  One process per measurement.
# Usage: ab_encode_variants.sh variants/libzbra_a.so variants/libzbra_b.so ...
root=$(cd "$(dirname "$0")/.." && pwd)
cd "$root"
for lib in default "$@"; do
    if [ "$lib" = default ]; then unset ZBRA_HIP_LIB; else export ZBRA_HIP_LIB=$root/$lib; fi
    echo "== $lib"
    for bits in 12 24 40 64; do
        timeout -k 10 120 python3 scripts/enc_width_prof.py $bits 27 10 2>/dev/null | tail -1 || exit 1
    done
    timeout -k 10 200 python3 scripts/bench_configs.py --reps 5 2>/dev/null | python3 -c "
import json, sys
for ln in sys.stdin:
    try: d = json.loads(ln)
    except Exception: continue
    print('  ', {k: v for k, v in d.items() if k in ('config', 'label', 'encode_ms', 'decode_ms', 'what')})
" || exit 1
done
