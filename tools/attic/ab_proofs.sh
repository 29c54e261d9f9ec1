#!/bin/bash
# A/B of two builds on the same box: tools/ab/libdehalo_prev.so (DEHALO_LIBRARY) against the tree's; proofs at k = 17 / 11 / 20 and batch mode
for round in 1 2; do
for lib in prev new; do
  if [ $lib = prev ]; then export DEHALO_LIBRARY=$PWD/tools/ab/libdehalo_prev.so; else unset DEHALO_LIBRARY; fi
  echo "== $lib (round $round)"
  python3 tools/profile_native_proof.py 17 delay_enc 60 2>/dev/null | grep "k = 17"
  python3 tools/profile_native_proof.py 11 pose_enc 60 2>/dev/null | grep "k = 11"
  [ $round = 1 ] && python3 tools/profile_native_proof.py 20 delay_enc 6 2>/dev/null | grep "k = 20"
  python3 tools/batch_trace.py 17 4 64 0 1 2>/dev/null | grep batch
  python3 tools/msm_small.py 20 uniform 1 2>&1 | tail -1
done; done
