#!/bin/bash
# batch mode with every prover's stream confined to a share of the compute units (DEHALO_CU_PARTITION = P: the i-th context gets the (i mod P)-th P-th of the CUs;
# DEHALO_CU_PARTITION_INTERLEAVE: CU c belongs to share c mod P instead of a contiguous block) against all provers on the whole chip
. tools/exp_lib.sh      # the switches below exist in the measurement build only (make EXPERIMENTS=1)
need_switch DEHALO_CU_PARTITION DEHALO_CU_PARTITION_INTERLEAVE 
run() { timeout -k 10 200 python tools/batch_trace.py 17 $1 64 0 1 2>/dev/null | grep batch; }
for round in 1 2; do
  echo "== whole chip, 4 provers (round $round)"; run 4
  echo "== 4 shares, contiguous, 4 provers"; DEHALO_CU_PARTITION=4 run 4
  echo "== 4 shares, interleaved, 4 provers"; DEHALO_CU_PARTITION=4 DEHALO_CU_PARTITION_INTERLEAVE=1 run 4
  echo "== 2 shares, contiguous, 4 provers"; DEHALO_CU_PARTITION=2 run 4
  echo "== 4 shares, contiguous, 8 provers"; DEHALO_CU_PARTITION=4 run 8
  echo "== 8 shares, contiguous, 8 provers"; DEHALO_CU_PARTITION=8 run 8
done
