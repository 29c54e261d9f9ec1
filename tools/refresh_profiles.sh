#!/bin/bash
# Regenerates the round's measurement artifacts on the GPU box into gpurun_out/final/ (copied into profiles/ afterwards).
# usage: tools/refresh_profiles.sh KERNEL_REV
set -e
rev=${1:?kernel revision}
export TMPDIR=/tmp
out=gpurun_out/final; rm -rf $out; mkdir -p $out
echo "[1] bench line"; timeout -k 10 600 python bench.py > $out/bench.json 2> $out/bench.err
echo "[2] kernel stats, timed region"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -o kt -- python3 bench.py --in-process --no-single-stream --no-cpu-baseline --proof-k 0 --proofs 0 > $out/bench_timed_region_under_rocprof.json 2> $out/kt.err
echo "[3] kernel stats, create_proof k=17"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kp -o kp -- python3 tools/profile_proof.py 17 1 10 > $out/create_proof_k17_phases_under_rocprof.txt 2> $out/kp.err
python3 tools/timeline.py $out/kp/kp_kernel_trace.csv > $out/create_proof_k17_kernel_timeline.txt; rm -f $out/kp/kp_kernel_trace.csv $out/kt/kt_kernel_trace.csv
echo "[3b] create_proof k=17, unprofiled"; timeout -k 10 300 python tools/profile_proof.py 17 1 10 > $out/create_proof_k17_phases.txt 2>/dev/null
echo "[4] pmc"; bash tools/collect_pmc.sh $rev > $out/pmc.log 2>&1; cp gpurun_out/pmc/pmc_traffic.json $out/; mkdir -p $out/pmc; cp gpurun_out/pmc/main/*.csv gpurun_out/pmc/calib/*.csv $out/pmc/ 2>/dev/null || true
for c in FETCH_SIZE WRITE_SIZE; do cp gpurun_out/pmc/main/${c}_counter_collection.csv $out/pmc/main_${c}.csv; cp gpurun_out/pmc/calib/${c}_counter_collection.csv $out/pmc/calib_${c}.csv; done
echo "[5] host paths"; timeout -k 10 300 python tools/host_path_bench.py > $out/host_path_measurements.txt 2>/dev/null
echo "[5b] field-vector primitives"; timeout -k 10 400 python tools/poly_bench.py > $out/field_vector_primitives.txt 2>/dev/null
echo "[6] one-shot sweep"; timeout -k 10 300 python tools/sweep_single_row.py pallas > $out/one_shot_window_sweep.txt 2>/dev/null
echo "[7] acc points sweep"; bash tools/sweep_acc_points.sh > $out/sweep_acc_points.txt 2>&1
echo "[8] two ranks on one GPU (gloo)"
timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --dist-backend gloo --force-device 0 --proofs 6 --no-cpu-baseline > $out/bench_2rank_one_gpu_gloo.log 2>&1 || echo "2-rank run failed"
rm -rf $out/kt/*agent_info* $out/kp/*agent_info*
ls -la $out
