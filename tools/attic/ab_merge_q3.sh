#!/bin/bash
# class 3 of k_msm_merge2 (9..64 records a bucket): the widest group that fits one sweep (default) against the first version's 8-or-1 choice (DEHALO_MSM_MERGE_Q3=1)
. tools/exp_lib.sh      # the switches below exist in the measurement build only (make EXPERIMENTS=1)
need_switch DEHALO_MSM_MERGE_Q3 DEHALO_MSM_MERGE_STAMPS 
for q in 1 0; do
  echo "== DEHALO_MSM_MERGE_Q3=$q: block stamps"
  for kb in "17 1" "14 3" "20 1" "11 1"; do set -- $kb; DEHALO_MSM_MERGE_Q3=$q DEHALO_MSM_MERGE_STAMPS=1 timeout -k 10 120 python tools/accum_eff.py $1 $2 2>&1 | grep "k_msm_merge2" | tail -1; done
  echo "== DEHALO_MSM_MERGE_Q3=$q: kernels by shape"
  tools/accum_eff.sh DEHALO_MSM_MERGE_Q3=$q
done
