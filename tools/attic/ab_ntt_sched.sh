#!/bin/bash
# NTT units compiled with alternative scheduling strategies (tools/ab/libdehalo_ntt_<strategy>.so) against the tree's build
for lib in tree iterative-ilp max-ilp tree; do
  if [ $lib = tree ]; then unset DEHALO_LIBRARY; else export DEHALO_LIBRARY=$PWD/tools/ab/libdehalo_ntt_$lib.so; fi
  echo "== $lib"
  [ $lib != tree ] && (python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "ntt or domain or coset" 2>&1 | tail -1)
  python3 tools/ntt_bench.py 2>/dev/null
  python3 tools/profile_native_proof.py 17 delay_enc 60 2>/dev/null | grep "k = 17"
done
