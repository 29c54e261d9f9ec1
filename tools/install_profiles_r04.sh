#!/bin/bash
# copies gpurun_out/final/* (tools/refresh_profiles_r04.sh a, b) into profiles/ under round 4's names
r=r04; F=gpurun_out/final
cp $F/bench.json profiles/${r}_bench.json
cp $F/bench_one_step_at_a_time_under_rocprof.json profiles/${r}_bench_timed_region_under_rocprof.json
cp $F/k1/k1_kernel_stats.csv profiles/${r}_kernel_stats_timed_region.csv
cp $F/accum0_launch_durations.txt profiles/${r}_accum0_launch_durations.txt
cp $F/bench_timed_region_under_rocprof.json profiles/${r}_bench_timed_region_4_in_flight_under_rocprof.json
cp $F/kt/kt_kernel_stats.csv profiles/${r}_kernel_stats_timed_region_4_in_flight.csv
cp $F/kp/kp_kernel_stats.csv profiles/${r}_kernel_stats_create_proof_k17.csv
[ -f $F/pmc_traffic.json ] && cp $F/pmc_traffic.json profiles/pmc_traffic.json
for f in step_concurrency_4_in_flight.txt create_proof_k17_phases_under_rocprof.txt create_proof_k17_kernel_timeline.txt create_proof_k17_device_idle.txt create_proof_k17_phases.txt create_proof_k17_host_timeline.txt \
         create_proof_k11_phases.txt create_proof_k11_host_timeline.txt batch_throughput_by_provers.txt batch_busy_fraction.txt sq_counters_step.txt ntt_sq_counters.txt ubench_qmem.txt ntt_bench.txt rccl_one_rank_test.log msm_kernels_by_shape.txt bred_phase_stamps.txt create_proof_k11_kernel_timeline.txt; do
  if [ -f $F/$f ]; then      # a file whose committed copy starts with a '#' header keeps it
    hdr=""; [ -f profiles/${r}_$f ] && hdr=$(grep '^#' profiles/${r}_$f)
    { [ -n "$hdr" ] && echo "$hdr"; grep -v 'amdgpu.ids' $F/$f | grep -v '^#'; } > profiles/${r}_$f.new && mv profiles/${r}_$f.new profiles/${r}_$f
  fi
done
[ -f $F/bench_gpus2_bare_command.json ] && cp $F/bench_gpus2_bare_command.json profiles/${r}_bench_gpus2_bare_command.json
[ -f $F/bench_gpus2_bare_command.err ] && grep -v 'socket.cpp\|amdgpu.ids' $F/bench_gpus2_bare_command.err | tail -40 > profiles/${r}_bench_gpus2_bare_command.log
ls profiles | grep $r
