#!/bin/bash
# Interleaved COLD A/B of the default decode tile kernel against the two-tiles-per-workgroup LDS-DMA
# shape (direct = 2), on the columns the shape was meant for: u32 tags, widths 7-12, the headline
# column, one wide width, and a batch.  Variant = tile:threads:nt:direct:order.
set -e
cd "$(dirname "$0")/.."
V="2048:512:3:0:2 2048:512:3:2:2 2048:512:3:2:0 2048:512:3:2:1 2048:512:2:2:2"
echo "== u32 tags, 2^29 rows"; timeout -k 10 200 python3 scripts/ab_decode_cold.py $V --tags --rows-log2 29 --rounds 8
for a in 7 8 10 12; do echo "== amplitude $a bits, 2^27 rows"; timeout -k 10 200 python3 scripts/ab_decode_cold.py $V --amp-bits $a --rows-log2 27 --rounds 8; done
echo "== headline column, 2^28 rows"; timeout -k 10 200 python3 scripts/ab_decode_cold.py $V --rows-log2 28 --rounds 10
echo "== amplitude 4 bits, 2^27 rows (warm-up route off in every variant here)"; timeout -k 10 200 python3 scripts/ab_decode_cold.py $V --amp-bits 4 --rows-log2 27 --rounds 8
echo "== amplitude 25 bits, 2^27 rows"; timeout -k 10 200 python3 scripts/ab_decode_cold.py $V --amp-bits 25 --rows-log2 27 --rounds 8
echo "== headline column as 256 arrays in one batch, 2^28 rows"; timeout -k 10 200 python3 scripts/ab_decode_cold.py $V --rows-log2 28 --arrays 256 --rounds 8
