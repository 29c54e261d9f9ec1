#!/bin/bash
# like tools/ab_env.sh, with the MSM + NTT step bench first: tools/ab_env_step.sh NAME VALUE_A VALUE_B ...
. tools/exp_lib.sh      # the switches below exist in the measurement build only (make EXPERIMENTS=1)
name=$1; shift
for round in 1 2; do
for v in "$@"; do
  if [ "$v" = unset ]; then unset $name; else export $name=$v; fi
  echo "== $name=$v (round $round)"
  python3 bench.py --in-process --no-cpu-baseline --proof-k 0 --proofs 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('step: %.1f Mpoints/s, %.4f ms per step; alone %.4f ms (accum0 %.4f)' % (d['value'], d['ms_per_step'], d['single_stream']['ms_per_step'], d['roofline']['avg_kernel_ms']))"
  python3 tools/profile_native_proof.py 17 delay_enc 60 2>/dev/null | grep "k = 17"
  python3 tools/profile_native_proof.py 11 pose_enc 60 2>/dev/null | grep "k = 11"
  python3 tools/batch_trace.py 17 4 64 0 1 2>/dev/null | grep batch
done; done
