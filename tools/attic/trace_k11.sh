#!/bin/bash
# kernel timeline of one pose_enc K = 11 proof (and k = 14 delay_enc) -> gpurun_out/trace_k11/
out=gpurun_out/trace_k11; rm -rf gpurun_out/trace_k11; mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $out/kp -o kp -- python3 tools/profile_native_proof.py 11 pose_enc 20 > $out/proof.log 2>&1
f=$(find $out/kp -name "kp_kernel_trace.csv" | head -1)
python3 tools/timeline.py $f 5 > $out/proof_timeline.txt
grep "k = 11" $out/proof.log
find $out -name "*kernel_trace.csv" -delete
