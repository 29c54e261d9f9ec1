#!/bin/bash
# copies gpurun_out/final6/* (tools/refresh_profiles_r06.sh a, b) into profiles/ under round 6's names
r=r06; F=gpurun_out/final6
[ -f $F/bench.json ] && cp $F/bench.json profiles/${r}_bench.json
[ -f $F/bench_full.json ] && cp $F/bench_full.json profiles/${r}_bench_verbose_record.json
[ -f $F/bench_one_step_at_a_time_under_rocprof.json ] && cp $F/bench_one_step_at_a_time_under_rocprof.json profiles/${r}_bench_timed_region_under_rocprof.json
[ -f $F/k1/k1_kernel_stats.csv ] && cp $F/k1/k1_kernel_stats.csv profiles/${r}_kernel_stats_timed_region.csv
[ -f $F/kp/kp_kernel_stats.csv ] && cp $F/kp/kp_kernel_stats.csv profiles/${r}_kernel_stats_create_proof_k17.csv
for f in accum0_launch_durations.txt create_proof_k17_phases_under_rocprof.txt create_proof_k17_kernel_timeline.txt create_proof_k17_device_idle.txt create_proof_k17_phases.txt create_proof_k17_host_timeline.txt synthesize_k17.txt; do
  [ -f $F/$f ] && grep -v 'amdgpu.ids' $F/$f > profiles/${r}_$f
done
[ -f $F/bench_gpus5_bare_command.json ] && tail -1 $F/bench_gpus5_bare_command.json > profiles/${r}_bench_gpus5_bare_command.json      # (gloo prints a connection notice on stdout before the line)
[ -f $F/bench_gpus5_bare_command.err ] && grep -v 'socket.cpp\|amdgpu.ids' $F/bench_gpus5_bare_command.err | tail -40 > profiles/${r}_bench_gpus5_bare_command.log
ls profiles | grep $r
