#!/bin/bash
# lookup tables of fixed columns handed to the permutation as distinct rows + multiplicities (default) against sorted in full every proof (DEHALO_PROVER_TABLE_ROWS=0)
. tools/exp_lib.sh      # the switches below exist in the measurement build only (make EXPERIMENTS=1)
need_switch DEHALO_PROVER_TABLE_ROWS 
for round in 1 2; do for v in 0 1; do
  export DEHALO_PROVER_TABLE_ROWS=$v
  echo "== DEHALO_PROVER_TABLE_ROWS=$v, round $round"
  timeout -k 10 200 python tools/profile_native_proof.py 17 delay_enc 60 2>/dev/null | grep "k = 17\|phases"
  timeout -k 10 200 python tools/profile_native_proof.py 17 mod_pow 40 2>/dev/null | grep "k = 17"
  timeout -k 10 200 python tools/batch_trace.py 17 4 64 0 1 2>/dev/null | grep batch
done; done
