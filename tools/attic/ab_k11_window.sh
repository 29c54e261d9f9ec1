#!/bin/bash
# pose_enc K = 11 and delay_enc k = 14 by the Pippenger window of their tables (DEHALO_WINDOW_BITS), two rounds
. tools/exp_lib.sh      # the switches below exist in the measurement build only (make EXPERIMENTS=1)
need_switch DEHALO_WINDOW_BITS 
for round in 1 2; do for c in unset 8 9 10 11 12; do
  if [ "$c" = unset ]; then unset DEHALO_WINDOW_BITS; else export DEHALO_WINDOW_BITS=$c; fi
  echo "== window $c (round $round)"
  python3 tools/profile_native_proof.py 11 pose_enc 60 2>/dev/null | grep "k = 11"
  [ $round = 1 ] && python3 tools/profile_native_proof.py 14 delay_enc 40 2>/dev/null | grep "k = 14"
done; done
