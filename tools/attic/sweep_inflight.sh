#!/bin/bash
# the MSM + NTT step by the number of steps in flight (each on its own context), two rounds
for round in 1 2; do for f in 2 3 4 5 6 8; do
  python3 bench.py --in-process --no-cpu-baseline --proof-k 0 --proofs 0 --inflight $f --steps 40 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('inflight $f (round $round): %.1f Mpoints/s, %.4f ms per step' % (d['value'], d['ms_per_step']))"
done; done
