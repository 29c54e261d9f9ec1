#!/bin/bash
# k_msm_accum0 (and the MSM's other kernels) by shape: tools/accum_eff.sh [ENV=VALUE ...]
for kv in "$@"; do export "$kv"; done
export TMPDIR=/tmp
out=gpurun_out/accum_eff; mkdir -p $out
for k in 17 14 20; do
  b="1,2,3,4,6"; [ $k = 20 ] && b="1,2"
  timeout -k 10 250 rocprofv3 --kernel-trace --output-format csv -d $out/t$k -o t -- python3 tools/accum_eff.py $k $b > $out/shapes_$k.jsonl 2> $out/err_$k.txt
  f=$(find $out/t$k -name "t_kernel_trace.csv" | head -1)
  python3 tools/accum_eff_read.py $out/shapes_$k.jsonl $f
  find $out/t$k -name "*.csv" -delete
done
