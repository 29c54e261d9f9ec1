#!/bin/bash
# copies gpurun_out/final/* (tools/refresh_profiles.sh) into profiles/ under the round's names.   usage: tools/install_profiles.sh r02
set -e
r=${1:?round prefix}; F=gpurun_out/final
cp $F/bench.json profiles/${r}_bench.json
cp $F/bench_timed_region_under_rocprof.json profiles/${r}_bench_timed_region_under_rocprof.json
cp $F/kt/kt_kernel_stats.csv profiles/${r}_kernel_stats_timed_region.csv
cp $F/kp/kp_kernel_stats.csv profiles/${r}_kernel_stats_create_proof_k17.csv
cp $F/create_proof_k17_phases_under_rocprof.txt profiles/${r}_create_proof_k17_phases_under_rocprof.txt
cp $F/create_proof_k17_phases.txt profiles/${r}_create_proof_k17_phases.txt
cp $F/create_proof_k17_kernel_timeline.txt profiles/${r}_create_proof_k17_kernel_timeline.txt
cp $F/pmc_traffic.json profiles/pmc_traffic.json
mkdir -p profiles/${r}_pmc
cp $F/pmc/main_FETCH_SIZE.csv profiles/${r}_pmc/fetch_counter_collection.csv
cp $F/pmc/main_WRITE_SIZE.csv profiles/${r}_pmc/write_counter_collection.csv
cp $F/pmc/calib_FETCH_SIZE.csv profiles/${r}_pmc/calib_fetch_counter_collection.csv
cp $F/pmc/calib_WRITE_SIZE.csv profiles/${r}_pmc/calib_write_counter_collection.csv
cp $F/host_path_measurements.txt profiles/${r}_host_path_measurements.txt
[ -f $F/field_vector_primitives.txt ] && cp $F/field_vector_primitives.txt profiles/${r}_field_vector_primitives.txt
cp $F/one_shot_window_sweep.txt profiles/${r}_one_shot_window_sweep.txt
cp $F/sweep_acc_points.txt profiles/${r}_sweep_acc_points.txt
grep -v 'socket.cpp\|amdgpu.ids' $F/bench_2rank_one_gpu_gloo.log > profiles/${r}_bench_2rank_one_gpu_gloo.log
