#!/bin/bash
# copies gpurun_out/final/* (tools/refresh_profiles_r05.sh a, b, c) into profiles/ under round 5's names
r=r05; F=gpurun_out/final
[ -f $F/bench.json ] && cp $F/bench.json profiles/${r}_bench.json
[ -f $F/bench_full.json ] && cp $F/bench_full.json profiles/${r}_bench_verbose_record.json
[ -f $F/bench_one_step_at_a_time_under_rocprof.json ] && cp $F/bench_one_step_at_a_time_under_rocprof.json profiles/${r}_bench_timed_region_under_rocprof.json
[ -f $F/k1/k1_kernel_stats.csv ] && cp $F/k1/k1_kernel_stats.csv profiles/${r}_kernel_stats_timed_region.csv
[ -f $F/kp/kp_kernel_stats.csv ] && cp $F/kp/kp_kernel_stats.csv profiles/${r}_kernel_stats_create_proof_k17.csv
[ -f $F/k20/k20_kernel_stats.csv ] && cp $F/k20/k20_kernel_stats.csv profiles/${r}_kernel_stats_create_proof_k20.csv
for f in accum0_launch_durations.txt create_proof_k17_phases_under_rocprof.txt create_proof_k17_kernel_timeline.txt create_proof_k17_device_idle.txt create_proof_k17_phases.txt create_proof_k17_host_timeline.txt \
         create_proof_k11_phases.txt create_proof_k11_host_timeline.txt create_proof_k20_phases.txt create_proof_k20_host_timeline.txt create_proof_k20_kernel_timeline.txt create_proof_k20_device_idle.txt \
         create_proof_k20_phases_under_rocprof.txt batch_throughput_by_provers.txt ntt_sq_counters.txt ntt_pass_durations_without_counters.txt ntt_bench.txt bred_phase_stamps.txt; do
  if [ -f $F/$f ]; then      # a file whose committed copy starts with a '#' header keeps it
    hdr=""; [ -f profiles/${r}_$f ] && hdr=$(grep '^#' profiles/${r}_$f)
    { [ -n "$hdr" ] && echo "$hdr"; grep -v 'amdgpu.ids' $F/$f | grep -v '^#'; } > profiles/${r}_$f.new && mv profiles/${r}_$f.new profiles/${r}_$f
  fi
done
[ -f $F/bench_gpus2_bare_command.json ] && tail -1 $F/bench_gpus2_bare_command.json > profiles/${r}_bench_gpus2_bare_command.json      # (gloo prints a connection notice on stdout before the line)
[ -f $F/bench_gpus2_bare_command.err ] && grep -v 'socket.cpp\|amdgpu.ids' $F/bench_gpus2_bare_command.err | tail -40 > profiles/${r}_bench_gpus2_bare_command.log
ls profiles | grep $r
