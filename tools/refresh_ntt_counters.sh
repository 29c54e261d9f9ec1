#!/bin/bash
# round 5: SQ counters of k_ntt_pass with the clocks up (two --pmc passes with --kernel-trace only), and the same workload's launch durations without counters
export TMPDIR=/tmp
out=gpurun_out/final; mkdir -p $out
for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU"; do
  tag=$(echo $pass | cut -d' ' -f1)
  rm -rf $out/sq_ntt_$tag
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $out/sq_ntt_$tag -- python3 tools/ntt_pmc.py > $out/sq_ntt_$tag.log 2>&1
done
python3 tools/sq_summary.py $out/sq_ntt_SQ_WAVE_CYCLES $out/sq_ntt_SQ_ACTIVE_INST_ANY --group-by-launch 3 --skip-launches 180 > $out/ntt_sq_counters.txt
rm -rf $out/kn
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kn -o kn -- python3 tools/ntt_pmc.py > $out/kn.log 2>&1
python3 tools/accum0_launches.py $out/kn/kn_kernel_trace.csv k_ntt_pass 30 > $out/ntt_pass_durations_without_counters.txt
cat $out/ntt_sq_counters.txt $out/ntt_pass_durations_without_counters.txt
find $out -name "*kernel_trace.csv" -delete; find $out -name "*agent_info*" -delete
