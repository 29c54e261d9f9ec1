#!/bin/bash
# small MSM launches: sort blocks of 256 scalars (DEHALO_MSM_SMALL_SLICES, default on) and k_msm_bucket blocks of fewer slices (DEHALO_MSM_BUCKET_FILL, default on) against the fixed
# 2048 scalars / 4 slices: kernels by shape at 2^11 and 2^14, then K = 11 and k = 14 proofs
. tools/exp_lib.sh      # the switches below exist in the measurement build only (make EXPERIMENTS=1)
need_switch DEHALO_MSM_BUCKET_FILL DEHALO_MSM_SMALL_SLICES 
export TMPDIR=/tmp
out=gpurun_out/small_sort; mkdir -p $out
for v in 0 1; do
  export DEHALO_MSM_SMALL_SLICES=$v DEHALO_MSM_BUCKET_FILL=$v
  echo "== DEHALO_MSM_SMALL_SLICES=$v DEHALO_MSM_BUCKET_FILL=$v"
  for k in 11 14; do
    timeout -k 10 250 rocprofv3 --kernel-trace --output-format csv -d $out/t$k -o t -- python3 tools/accum_eff.py $k 1,2,5 > $out/shapes_$k.jsonl 2> $out/err_$k.txt
    f=$(find $out/t$k -name "t_kernel_trace.csv" | head -1)
    python3 tools/accum_eff_read.py $out/shapes_$k.jsonl $f
    find $out/t$k -name "*.csv" -delete
  done
  for r in 1 2; do
    timeout -k 10 200 python tools/profile_native_proof.py 11 pose_enc 60 2>/dev/null | grep "k = 11"
    timeout -k 10 200 python tools/profile_native_proof.py 16 mod_pow 40 2>/dev/null | grep "k = 16"
  done
done
