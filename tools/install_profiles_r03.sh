#!/bin/bash
# copies gpurun_out/final/* (tools/refresh_profiles_r03.sh) into profiles/ under round 3's names
set -e
r=r03; F=gpurun_out/final
cp $F/bench.json profiles/${r}_bench.json
cp $F/bench_one_step_at_a_time_under_rocprof.json profiles/${r}_bench_timed_region_under_rocprof.json
cp $F/k1/k1_kernel_stats.csv profiles/${r}_kernel_stats_timed_region.csv
cp $F/accum0_launch_durations.txt profiles/${r}_accum0_launch_durations.txt
cp $F/bench_timed_region_under_rocprof.json profiles/${r}_bench_timed_region_4_in_flight_under_rocprof.json
cp $F/kt/kt_kernel_stats.csv profiles/${r}_kernel_stats_timed_region_4_in_flight.csv
cp $F/kp/kp_kernel_stats.csv profiles/${r}_kernel_stats_create_proof_k17.csv
for f in create_proof_k17_phases_under_rocprof.txt create_proof_k17_kernel_timeline.txt create_proof_k17_device_idle.txt create_proof_k17_phases.txt create_proof_k17_host_timeline.txt create_proof_k11_phases.txt create_proof_k11_host_timeline.txt \
         batch_throughput_by_provers.txt batch_busy_fraction.txt native_vs_python_k17.txt stream_concurrency.txt ubench_mfma_price.txt pmc_calib.txt ubench_chain.txt ntt_bench.txt clock_ramp.txt; do
  grep -v 'amdgpu.ids' $F/$f > profiles/${r}_$f || true
done
grep -v 'socket.cpp\|amdgpu.ids' $F/bench_2rank_one_gpu_gloo.log > profiles/${r}_bench_2rank_one_gpu_gloo.log
