#!/bin/bash
# Regenerates round 6's measurement artifacts on the GPU box into gpurun_out/final6/ (tools/install_profiles_r06.sh copies them into profiles/).
# Two parts, each within one gpurun call:  tools/refresh_profiles_r06.sh a | b
export TMPDIR=/tmp
out=gpurun_out/final6
part=${1:-a}
mkdir -p $out
if [ "$part" = a ]; then
echo "[1] bench line (the driver's command) + the verbose record"; timeout -k 10 900 python bench.py --full-out $out/bench_full.json > $out/bench.json 2> $out/bench.err || echo "bench failed"
wc -c $out/bench.json
echo "[2] kernel stats, timed region one step at a time (the duration the roofline is computed from)"
rm -rf $out/k1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/k1 -o k1 -- python3 bench.py --in-process --inflight 1 --no-single-stream --no-cpu-baseline --proof-k 0 --proofs 0 --full-out "" > $out/bench_one_step_at_a_time_under_rocprof.json 2> $out/k1.err
python3 tools/accum0_launches.py $out/k1/k1_kernel_trace.csv > $out/accum0_launch_durations.txt
echo "[3] kernel stats, dehalo_create_proof k=17 on the row-matched witness"
rm -rf $out/kp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kp -o kp -- python3 tools/profile_native_proof.py 17 delay_enc 40 > $out/create_proof_k17_phases_under_rocprof.txt 2> $out/kp.err
python3 tools/timeline.py $out/kp/kp_kernel_trace.csv > $out/create_proof_k17_kernel_timeline.txt
python3 tools/proof_gaps.py $out/kp/kp_kernel_trace.csv > $out/create_proof_k17_device_idle.txt
timeout -k 10 300 python tools/profile_native_proof.py 17 delay_enc 40 > $out/create_proof_k17_phases.txt 2> $out/create_proof_k17_host_timeline.txt
find $out -name "*kernel_trace.csv" -delete
find $out -name "*agent_info*" -delete
else
echo "[4] N = 5 from the bare command (five ranks on this one GPU, gloo for the gather: the pool allows six processes on a card and the launcher is one of them): both batches, every proof re-made alone"
timeout -k 10 1000 python3 bench.py --gpus 5 --dist-backend gloo --force-device 0 --proofs 10 --fixed-batch 64 --no-cpu-baseline --full-out "" > $out/bench_gpus5_bare_command.json 2> $out/bench_gpus5_bare_command.err; echo "rc=$?" >> $out/bench_gpus5_bare_command.err
echo "[5] synthesize at the metric's size"
python tools/synth_bench.py > $out/synthesize_k17.txt 2>&1
fi
