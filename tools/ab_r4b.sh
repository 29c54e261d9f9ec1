#!/bin/bash
# round 4 A/B, second set: round-3 tails against the new ones, the batched GraphEvaluator launches, the window size with the new tails
run() {
  echo "== $1"
  python3 bench.py --in-process --no-cpu-baseline --proof-k 0 --proofs 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('step: %.1f Mpoints/s, %.4f ms per step; alone %.4f ms %s' % (d['value'], d['ms_per_step'], d['single_stream']['ms_per_step'], d['single_stream']['kernel_ms']))"
  python3 tools/profile_native_proof.py 17 delay_enc 60 2>/dev/null | grep "k = 17"
  python3 tools/profile_native_proof.py 11 pose_enc 60 2>/dev/null | grep "k = 11"
  python3 tools/batch_trace.py 17 4 64 0 1 2>/dev/null | grep batch
}
for round in 1 2; do
  ( export DEHALO_MSM_BRED=0 DEHALO_MSM_MERGE2=0 DEHALO_GRAPH_BATCH=0; run "round-3 kernels (BRED=0 MERGE2=0 GRAPH_BATCH=0), round $round" )
  ( run "round-4 defaults, round $round" )
  ( export DEHALO_GRAPH_BATCH=0; run "round-4 tails, GraphEvaluator launches one by one, round $round" )
  ( export DEHALO_WINDOW_BITS=16; run "round-4 defaults, window 16 for every table, round $round" )
  ( export DEHALO_WINDOW_BITS=14; run "round-4 defaults, window 14 for every table, round $round" )
done
