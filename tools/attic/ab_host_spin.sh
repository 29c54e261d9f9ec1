#!/bin/bash
# host waits that poll before they block (DEHALO_HOST_SPIN_US) against the runtime's blocking wait (0): K = 11, k = 17 proofs and the 64-proof batch, two rounds
. tools/exp_lib.sh      # the switches below exist in the measurement build only (make EXPERIMENTS=1)
need_switch DEHALO_HOST_SPIN_US 
for round in 1 2; do for v in 0 300 5000; do
  export DEHALO_HOST_SPIN_US=$v
  echo "== DEHALO_HOST_SPIN_US=$v, round $round"
  timeout -k 10 200 python tools/profile_native_proof.py 11 pose_enc 60 2>/dev/null | grep "k = 11"
  timeout -k 10 200 python tools/profile_native_proof.py 17 delay_enc 60 2>/dev/null | grep "k = 17"
  timeout -k 10 200 python tools/batch_trace.py 17 4 64 0 1 2>/dev/null | grep batch
done; done
