#!/bin/bash
# tools/ab_env_steponly.sh NAME VALUE_A VALUE_B ...: the MSM + NTT step bench only, four alternating rounds
. tools/exp_lib.sh      # the switches below exist in the measurement build only (make EXPERIMENTS=1)
name=$1; shift
for round in 1 2 3 4; do for v in "$@"; do
  if [ "$v" = unset ]; then unset $name; else export $name=$v; fi
  python3 bench.py --in-process --no-cpu-baseline --proof-k 0 --proofs 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$name=$v round $round: %.1f Mpoints/s, %.4f ms per step; alone %.4f ms (accum0 %.4f)' % (d['value'], d['ms_per_step'], d['single_stream']['ms_per_step'], d['roofline']['avg_kernel_ms']))"
done; done
