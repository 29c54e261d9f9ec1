#!/bin/bash
# the fewest layers (waves per SIMD) of the accumulation's grid: 4 (a full chip whatever the size) against 3, 2, 1 -- fewer lanes with longer ranges leave fewer partial sums to merge
. tools/exp_lib.sh      # the switches below exist in the measurement build only (make EXPERIMENTS=1)
need_switch DEHALO_MSM_ACC_MIN_LAYERS 
for v in 4 2 1; do echo "== DEHALO_MSM_ACC_MIN_LAYERS=$v: kernels by shape"; tools/accum_eff.sh DEHALO_MSM_ACC_MIN_LAYERS=$v; done
for v in 4 3 2 1; do echo "== DEHALO_MSM_ACC_MIN_LAYERS=$v"; tools/ab_quick.sh DEHALO_MSM_ACC_MIN_LAYERS=$v; done
