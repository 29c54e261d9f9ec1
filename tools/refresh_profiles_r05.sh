#!/bin/bash
# Regenerates round 5's measurement artifacts on the GPU box into gpurun_out/final/ (tools/install_profiles_r05.sh copies them into profiles/).
# Three parts, each within one gpurun call:  tools/refresh_profiles_r05.sh a | b | c
export TMPDIR=/tmp
out=gpurun_out/final
part=${1:-a}
mkdir -p $out
if [ "$part" = a ]; then
echo "[1] bench line (the driver's command) + the verbose record"; timeout -k 10 900 python bench.py --full-out $out/bench_full.json > $out/bench.json 2> $out/bench.err || echo "bench failed"
wc -c $out/bench.json
echo "[2] kernel stats, timed region one step at a time (the duration the roofline is computed from)"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/k1 -o k1 -- python3 bench.py --in-process --inflight 1 --no-single-stream --no-cpu-baseline --proof-k 0 --proofs 0 --full-out "" > $out/bench_one_step_at_a_time_under_rocprof.json 2> $out/k1.err
python3 tools/accum0_launches.py $out/k1/k1_kernel_trace.csv > $out/accum0_launch_durations.txt
echo "[3] kernel stats, dehalo_create_proof k=17"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kp -o kp -- python3 tools/profile_native_proof.py 17 delay_enc 40 > $out/create_proof_k17_phases_under_rocprof.txt 2> $out/kp.err
python3 tools/timeline.py $out/kp/kp_kernel_trace.csv > $out/create_proof_k17_kernel_timeline.txt
python3 tools/proof_gaps.py $out/kp/kp_kernel_trace.csv > $out/create_proof_k17_device_idle.txt
echo "[3b] dehalo_create_proof k=17, K=11, k=14 unprofiled"
timeout -k 10 300 python tools/profile_native_proof.py 17 delay_enc 40 > $out/create_proof_k17_phases.txt 2> $out/create_proof_k17_host_timeline.txt
timeout -k 10 300 python tools/profile_native_proof.py 11 pose_enc 40 > $out/create_proof_k11_phases.txt 2> $out/create_proof_k11_host_timeline.txt
echo "[4] k = 20: phases, host timeline, kernel timeline, kernel stats"
timeout -k 10 400 python tools/profile_native_proof.py 20 delay_enc 10 > $out/create_proof_k20_phases.txt 2> $out/create_proof_k20_host_timeline.txt
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $out/k20 -o k20 -- python3 tools/profile_native_proof.py 20 delay_enc 6 > $out/create_proof_k20_phases_under_rocprof.txt 2> $out/k20.err
python3 tools/timeline.py $out/k20/k20_kernel_trace.csv > $out/create_proof_k20_kernel_timeline.txt
python3 tools/proof_gaps.py $out/k20/k20_kernel_trace.csv > $out/create_proof_k20_device_idle.txt
find gpurun_out/final -name "*kernel_trace.csv" -delete
find gpurun_out/final -name "*agent_info*" -delete
elif [ "$part" = b ]; then
echo "[5] SQ counters: k_ntt_pass"
WARM=${WARM:-60}      # untimed launches of ntt_pmc.py (NTT_PMC_WARM): the counters' window starts 3 passes x WARM launches in
export NTT_PMC_WARM=$WARM
for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU"; do
  tag=$(echo $pass | cut -d' ' -f1)
  rm -rf $out/sq_ntt_$tag      # (a re-run must not leave an earlier run's CSVs for sq_summary.py to pick up)
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $out/sq_ntt_$tag -- python3 tools/ntt_pmc.py > $out/sq_ntt_$tag.log 2>&1
done
python3 tools/sq_summary.py $out/sq_ntt_SQ_WAVE_CYCLES $out/sq_ntt_SQ_ACTIVE_INST_ANY --group-by-launch 3 --skip-launches $((3 * WARM)) > $out/ntt_sq_counters.txt
rm -rf $out/kn
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kn -o kn -- python3 tools/ntt_pmc.py > $out/kn.log 2>&1
python3 tools/accum0_launches.py $out/kn/kn_kernel_trace.csv k_ntt_pass 30 > $out/ntt_pass_durations_without_counters.txt
echo "[6] batch mode: throughput by provers"
for p in 2 3 4 6; do timeout -k 10 200 python tools/batch_trace.py 17 $p 64 0 1 | grep batch; done > $out/batch_throughput_by_provers.txt 2>/dev/null
echo "[7] ntt bench; N = 2 from the bare command (two ranks on this one GPU, gloo for the gather)"
timeout -k 10 200 python tools/ntt_bench.py > $out/ntt_bench.txt 2>/dev/null
timeout -k 10 500 python3 bench.py --gpus 2 --dist-backend gloo --force-device 0 --proofs 8 --no-cpu-baseline --full-out "" > $out/bench_gpus2_bare_command.json 2> $out/bench_gpus2_bare_command.err; echo "rc=$?" >> $out/bench_gpus2_bare_command.err
find gpurun_out/final -name "*kernel_trace.csv" -delete
find gpurun_out/final -name "*agent_info*" -delete
else
echo "[8] measurement build (make EXPERIMENTS=1): the bucket reduction's phases and hand-offs"
mkdir -p gpurun_out/ab/exp
make -j16 gpurun_out/ab/exp/libdehalo.so LIB=gpurun_out/ab/exp/libdehalo.so OBJDIR=gpurun_out/ab/exp/obj EXPERIMENTS=1 > gpurun_out/ab/exp/build.log 2>&1 || { tail -20 gpurun_out/ab/exp/build.log; exit 1; }
export DEHALO_LIBRARY=$PWD/gpurun_out/ab/exp/libdehalo.so
for k in 11 17 20; do echo "== k $k"; DEHALO_MSM_BRED_STAMPS=1 timeout -k 10 120 python tools/accum_eff.py $k 1,4 2>&1 | grep "k_msm_bred" | awk "NR%4==0"; done > $out/bred_phase_stamps.txt 2>&1
fi
ls gpurun_out/final | head -80
