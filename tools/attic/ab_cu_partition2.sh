#!/bin/bash
# interleaved CU shares with several provers per share
. tools/exp_lib.sh      # the switches below exist in the measurement build only (make EXPERIMENTS=1)
need_switch DEHALO_CU_PARTITION DEHALO_CU_PARTITION_INTERLEAVE 
run() { timeout -k 10 200 python tools/batch_trace.py 17 $1 64 0 1 2>/dev/null | grep batch; }
export DEHALO_CU_PARTITION_INTERLEAVE=1
for round in 1 2; do
  echo "== whole chip, 4 provers (round $round)"; DEHALO_CU_PARTITION=0 run 4
  echo "== 4 shares interleaved, 4 provers"; DEHALO_CU_PARTITION=4 run 4
  echo "== 4 shares interleaved, 8 provers"; DEHALO_CU_PARTITION=4 run 8
  echo "== 4 shares interleaved, 12 provers"; DEHALO_CU_PARTITION=4 run 12
  echo "== 2 shares interleaved, 4 provers"; DEHALO_CU_PARTITION=2 run 4
  echo "== 2 shares interleaved, 6 provers"; DEHALO_CU_PARTITION=2 run 6
  echo "== 2 shares interleaved, 8 provers"; DEHALO_CU_PARTITION=2 run 8
  echo "== 3 shares interleaved, 6 provers"; DEHALO_CU_PARTITION=3 run 6
done
