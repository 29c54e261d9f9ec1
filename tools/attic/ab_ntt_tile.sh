#!/bin/bash
# half-size NTT tiles for small launches (DEHALO_NTT_SMALL_TILE_LOG): parity of the transforms with every multi-pass launch on small tiles, then timings
DEHALO_NTT_SMALL_TILE_LOG=40 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "ntt or domain or coset" 2>&1 | tail -2 || exit 1
for v in unset 20 21 22 40; do
  if [ "$v" = unset ]; then unset DEHALO_NTT_SMALL_TILE_LOG; else export DEHALO_NTT_SMALL_TILE_LOG=$v; fi
  echo "== DEHALO_NTT_SMALL_TILE_LOG=$v"; python3 tools/ntt_bench.py 2>/dev/null
done
