#!/bin/bash
# previous build (tools/ab/libdehalo_prev.so) against the tree's: MSM parity of the new one, then step bench, proofs, batch -- two rounds
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "msm" 2>&1 | tail -1 || exit 1
for round in 1 2; do for lib in prev new; do
  if [ $lib = prev ]; then export DEHALO_LIBRARY=$PWD/tools/ab/libdehalo_prev.so; else unset DEHALO_LIBRARY; fi
  echo "== $lib (round $round)"
  python3 bench.py --in-process --no-cpu-baseline --proof-k 0 --proofs 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('step: %.1f Mpoints/s, %.4f ms per step; alone %.4f ms (accum0 %.4f)' % (d['value'], d['ms_per_step'], d['single_stream']['ms_per_step'], d['roofline']['avg_kernel_ms']))"
  python3 tools/profile_native_proof.py 17 delay_enc 60 2>/dev/null | grep "k = 17"
  python3 tools/profile_native_proof.py 11 pose_enc 60 2>/dev/null | grep "k = 11"
  python3 tools/batch_trace.py 17 4 64 0 1 2>/dev/null | grep batch
done; done
