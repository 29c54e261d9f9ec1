#!/bin/bash
# A/B inside one build by an environment variable: tools/ab_env.sh NAME VALUE_A VALUE_B  (VALUE "unset" leaves it unset); proofs at k = 17 / 11 / 14 and batch mode
. tools/exp_lib.sh      # the switches below exist in the measurement build only (make EXPERIMENTS=1)
name=$1; shift
for round in 1 2; do
for v in "$@"; do
  if [ "$v" = unset ]; then unset $name; else export $name=$v; fi
  echo "== $name=$v (round $round)"
  python3 tools/profile_native_proof.py 17 delay_enc 60 2>/dev/null | grep "k = 17"
  python3 tools/profile_native_proof.py 11 pose_enc 60 2>/dev/null | grep "k = 11"
  [ $round = 1 ] && python3 tools/profile_native_proof.py 14 delay_enc 40 2>/dev/null | grep "k = 14"
  [ $round = 1 ] && python3 tools/profile_native_proof.py 20 delay_enc 5 2>/dev/null | grep "k = 20"
  python3 tools/batch_trace.py 17 4 64 0 1 2>/dev/null | grep batch
done; done
