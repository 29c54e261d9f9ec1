#!/bin/bash
# A/B of two builds of the library on the same box: tools/ab/libdehalo_prev.so (DEHALO_LIBRARY) against the tree's, MSM shapes of a proof + the proofs
for round in 1 2; do
for lib in prev new; do
  if [ $lib = prev ]; then export DEHALO_LIBRARY=$PWD/tools/ab/libdehalo_prev.so; else unset DEHALO_LIBRARY; fi
  echo "== $lib (round $round)"
  python3 tools/msm_small.py 17 uniform 1 2>&1 | tail -1
  python3 tools/msm_small.py 17 uniform 4 2>&1 | tail -1
  python3 tools/msm_small.py 17 witness 5 2>&1 | tail -1
  python3 tools/msm_small.py 17 lookup 10 2>&1 | tail -1
  python3 tools/msm_small.py 20 uniform 1 2>&1 | tail -1
  python3 tools/profile_native_proof.py 17 delay_enc 40 2>/dev/null | grep "k = 17"
  python3 tools/profile_native_proof.py 11 pose_enc 40 2>/dev/null | grep "k = 11"
done; done
