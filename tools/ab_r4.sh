#!/bin/bash
# round 4 A/B on one box: the bucket reduction (DEHALO_MSM_BRED), the merge (DEHALO_MSM_MERGE2) and the accumulation's block size (DEHALO_MSM_ACC_BLOCK)
# tools/ab_r4.sh [what=step,proof,batch] ; each configuration twice, alternating
what=${1:-step,proof,batch}
run() {
  echo "== $1"
  if [[ $what == *step* ]]; then
  python3 bench.py --in-process --no-cpu-baseline --proof-k 0 --proofs 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('step: %.1f Mpoints/s, %.4f ms per step; alone %.4f ms %s' % (d['value'], d['ms_per_step'], d['single_stream']['ms_per_step'], d['single_stream']['kernel_ms']))"
  fi
  if [[ $what == *proof* ]]; then
  python3 tools/profile_native_proof.py 17 delay_enc 60 2>/dev/null | grep "k = 17"
  python3 tools/profile_native_proof.py 11 pose_enc 60 2>/dev/null | grep "k = 11"
  fi
  if [[ $what == *batch* ]]; then
  python3 tools/batch_trace.py 17 4 64 0 1 2>/dev/null | grep batch
  fi
}
for round in 1 2; do
  ( export DEHALO_MSM_BRED=0 DEHALO_MSM_MERGE2=0; run "round-3 tails (BRED=0 MERGE2=0), round $round" )
  ( export DEHALO_MSM_MERGE2=0; run "bred, old merge, round $round" )
  ( run "bred + merge2, round $round" )
  ( export DEHALO_MSM_ACC_BLOCK=768; run "bred + merge2 + acc block 768, round $round" )
  ( export DEHALO_MSM_ACC_BLOCK=768 DEHALO_MSM_BRED=0 DEHALO_MSM_MERGE2=0; run "round-3 tails + acc block 768, round $round" )
done
