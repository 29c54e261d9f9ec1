#!/bin/bash
# batch mode and single proofs by the block size of the two scalar-decoding sort kernels (their LDS footprint decides what can sit beside them on a CU)
. tools/exp_lib.sh      # the switches below exist in the measurement build only (make EXPERIMENTS=1)
need_switch DEHALO_MSM_HIST_THREADS DEHALO_MSM_PART_THREADS DEHALO_NTT_SMALL_TILE_LOG 
run() { echo "== $1"; python3 tools/profile_native_proof.py 17 delay_enc 60 2>/dev/null | grep "k = 17"; python3 tools/batch_trace.py 17 4 64 0 1 2>/dev/null | grep batch; }
for round in 1 2; do
  ( run "default (1024 / 1024), round $round" )
  ( export DEHALO_MSM_PART_THREADS=512; run "part 512, round $round" )
  ( export DEHALO_MSM_PART_THREADS=256; run "part 256, round $round" )
  ( export DEHALO_MSM_PART_THREADS=512 DEHALO_MSM_HIST_THREADS=512; run "part 512 hist 512, round $round" )
  ( export DEHALO_NTT_SMALL_TILE_LOG=30; run "NTT half tiles everywhere, round $round" )
  ( export DEHALO_NTT_SMALL_TILE_LOG=30 DEHALO_MSM_PART_THREADS=512; run "NTT half tiles + part 512, round $round" )
done
