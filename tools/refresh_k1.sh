#!/bin/bash
# step [2] of tools/refresh_profiles_r03.sh alone: the one-step-at-a-time run under rocprofv3, its stats and the per-launch durations (into gpurun_out/final/)
export TMPDIR=/tmp
out=gpurun_out/final; mkdir -p $out; rm -rf $out/k1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/k1 -o k1 -- python3 bench.py --in-process --inflight 1 --no-single-stream --no-cpu-baseline --proof-k 0 --proofs 0 > $out/bench_one_step_at_a_time_under_rocprof.json 2> $out/k1.err
python3 tools/accum0_launches.py $out/k1/k1_kernel_trace.csv > $out/accum0_launch_durations.txt
rm -f $out/k1/k1_kernel_trace.csv; rm -rf $out/k1/*agent_info*
cat $out/accum0_launch_durations.txt
