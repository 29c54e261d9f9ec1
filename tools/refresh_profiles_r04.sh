#!/bin/bash
# Regenerates round 4's measurement artifacts on the GPU box into gpurun_out/final/ (tools/install_profiles_r04.sh copies them into profiles/).
# Two parts, each within one gpurun call:  tools/refresh_profiles_r04.sh a | b
export TMPDIR=/tmp
out=gpurun_out/final
part=${1:-a}
if [ "$part" = a ]; then
rm -rf gpurun_out/final
mkdir -p gpurun_out/final
echo "[1] bench line"; timeout -k 10 500 python bench.py > $out/bench.json 2> $out/bench.err || echo "bench failed"
echo "[2] kernel stats, timed region one step at a time (the duration the roofline is computed from)"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/k1 -o k1 -- python3 bench.py --in-process --inflight 1 --no-single-stream --no-cpu-baseline --proof-k 0 --proofs 0 > $out/bench_one_step_at_a_time_under_rocprof.json 2> $out/k1.err
python3 tools/accum0_launches.py $out/k1/k1_kernel_trace.csv > $out/accum0_launch_durations.txt
echo "[2b] kernel stats, timed region 4 steps in flight"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -o kt -- python3 bench.py --in-process --no-single-stream --no-cpu-baseline --proof-k 0 --proofs 0 > $out/bench_timed_region_under_rocprof.json 2> $out/kt.err
python3 tools/step_concurrency.py $out/kt/kt_kernel_trace.csv > $out/step_concurrency_4_in_flight.txt
echo "[3] kernel stats, dehalo_create_proof k=17"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kp -o kp -- python3 tools/profile_native_proof.py 17 delay_enc 40 > $out/create_proof_k17_phases_under_rocprof.txt 2> $out/kp.err
python3 tools/timeline.py $out/kp/kp_kernel_trace.csv > $out/create_proof_k17_kernel_timeline.txt
python3 tools/proof_gaps.py $out/kp/kp_kernel_trace.csv > $out/create_proof_k17_device_idle.txt
echo "[3b] dehalo_create_proof k=17 and K=11, unprofiled"
timeout -k 10 300 python tools/profile_native_proof.py 17 delay_enc 40 > $out/create_proof_k17_phases.txt 2> $out/create_proof_k17_host_timeline.txt
timeout -k 10 300 python tools/profile_native_proof.py 11 pose_enc 40 > $out/create_proof_k11_phases.txt 2> $out/create_proof_k11_host_timeline.txt
echo "[4] batch mode: throughput by provers, busy fraction"
for p in 2 3 4 6 8; do timeout -k 10 200 python tools/batch_trace.py 17 $p 64 0 1 | grep batch; done > $out/batch_throughput_by_provers.txt 2>/dev/null
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $out/bt -o bt -- python3 tools/batch_trace.py 17 4 32 0 1 > $out/bt.log 2>/dev/null
python3 tools/busy_fraction.py $out/bt/bt_kernel_trace.csv 32 > $out/batch_busy_fraction.txt
find gpurun_out/final -name "*kernel_trace.csv" -delete
find gpurun_out/final -name "*agent_info*" -delete
else
mkdir -p gpurun_out/final
echo "[5] HBM traffic of k_msm_accum0 (separate --pmc passes)"; bash tools/collect_pmc.sh r04a > $out/collect_pmc.log 2>&1; cp gpurun_out/pmc/pmc_traffic.json $out/pmc_traffic.json
echo "[6] SQ counters: the step's kernels one step at a time, and k_ntt_pass"
for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU"; do
  tag=$(echo $pass | cut -d' ' -f1)
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $out/sq_step_$tag -- python3 bench.py --in-process --no-cpu-baseline --proof-k 0 --proofs 0 --steps 5 --warmup 2 --preheat-s 0 --inflight 1 --no-single-stream > $out/sq_step_$tag.log 2>&1
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $out/sq_ntt_$tag -- python3 tools/ntt_pmc.py > $out/sq_ntt_$tag.log 2>&1
done
python3 tools/sq_summary.py $out/sq_step_SQ_WAVE_CYCLES $out/sq_step_SQ_ACTIVE_INST_ANY > $out/sq_counters_step.txt
python3 tools/sq_summary.py $out/sq_ntt_SQ_WAVE_CYCLES $out/sq_ntt_SQ_ACTIVE_INST_ANY --group-by-launch 3 > $out/ntt_sq_counters.txt
echo "[6b] the MSM's kernels by shape, the bucket reduction's phases, a K = 11 kernel timeline"
tools/accum_eff.sh > $out/msm_kernels_by_shape.txt 2>&1
for k in 11 17 20; do echo "== k $k"; DEHALO_MSM_BRED_STAMPS=1 timeout -k 10 120 python tools/accum_eff.py $k 1,4 2>&1 | grep "k_msm_bred" | awk "NR%4==0"; done > $out/bred_phase_stamps.txt 2>&1
tools/trace_k11.sh > $out/trace_k11.log 2>&1; cp gpurun_out/trace_k11/proof_timeline.txt $out/create_proof_k11_kernel_timeline.txt
echo "[7] microbenchmarks"; timeout -k 5 100 ./tools/ubench_qmem > $out/ubench_qmem.txt 2>&1; timeout -k 10 200 python tools/ntt_bench.py > $out/ntt_bench.txt 2>/dev/null
echo "[8] N = 2 from the bare command (two ranks on this one GPU, gloo for the gather), and the one-rank RCCL test"
timeout -k 10 400 python3 bench.py --gpus 2 --dist-backend gloo --force-device 0 --proofs 8 > $out/bench_gpus2_bare_command.json 2> $out/bench_gpus2_bare_command.err; echo "rc=$?" >> $out/bench_gpus2_bare_command.err
timeout -k 10 300 python -m pytest tests/test_sharding.py -q -m gpu -s > $out/rccl_one_rank_test.log 2>&1
find gpurun_out/final -name "*kernel_trace.csv" -delete
find gpurun_out/final -name "*agent_info*" -delete
fi
ls gpurun_out/final | head -60
