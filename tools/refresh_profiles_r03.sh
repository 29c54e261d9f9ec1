#!/bin/bash
# Regenerates round 3's measurement artifacts on the GPU box into gpurun_out/final/ (tools/install_profiles_r03.sh copies them into profiles/).
export TMPDIR=/tmp
out=gpurun_out/final; rm -rf $out; mkdir -p $out
echo "[1] bench line"; timeout -k 10 700 python bench.py > $out/bench.json 2> $out/bench.err || echo "bench failed"
echo "[2] kernel stats, timed region one step at a time (the duration the roofline is computed from)"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/k1 -o k1 -- python3 bench.py --in-process --inflight 1 --no-single-stream --no-cpu-baseline --proof-k 0 --proofs 0 > $out/bench_one_step_at_a_time_under_rocprof.json 2> $out/k1.err
python3 tools/accum0_launches.py $out/k1/k1_kernel_trace.csv > $out/accum0_launch_durations.txt
echo "[2b] kernel stats, timed region 4 steps in flight"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -o kt -- python3 bench.py --in-process --no-single-stream --no-cpu-baseline --proof-k 0 --proofs 0 > $out/bench_timed_region_under_rocprof.json 2> $out/kt.err
echo "[3] kernel stats, dehalo_create_proof k=17"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kp -o kp -- python3 tools/profile_native_proof.py 17 delay_enc 40 > $out/create_proof_k17_phases_under_rocprof.txt 2> $out/kp.err
python3 tools/timeline.py $out/kp/kp_kernel_trace.csv > $out/create_proof_k17_kernel_timeline.txt
python3 tools/proof_gaps.py $out/kp/kp_kernel_trace.csv > $out/create_proof_k17_device_idle.txt
echo "[3b] dehalo_create_proof k=17 and K=11, unprofiled"
timeout -k 10 300 python tools/profile_native_proof.py 17 delay_enc 40 > $out/create_proof_k17_phases.txt 2> $out/create_proof_k17_host_timeline.txt
timeout -k 10 300 python tools/profile_native_proof.py 11 pose_enc 40 > $out/create_proof_k11_phases.txt 2> $out/create_proof_k11_host_timeline.txt
echo "[4] batch mode: throughput by provers, busy fraction"
for p in 2 3 4 6 8; do timeout -k 10 200 python tools/batch_trace.py 17 $p 64 0 1 | grep batch; done > $out/batch_throughput_by_provers.txt 2>/dev/null
timeout -k 10 200 python tools/batch_trace.py 17 4 64 0 0 | grep batch >> $out/batch_throughput_by_provers.txt 2>/dev/null
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $out/bt -o bt -- python3 tools/batch_trace.py 17 4 32 0 1 > $out/bt.log 2>/dev/null
python3 tools/busy_fraction.py $out/bt/bt_kernel_trace.csv 32 > $out/batch_busy_fraction.txt
echo "[5] python-driven vs native prover"; timeout -k 10 400 python tools/native_bench.py 17 delay_enc 32 > $out/native_vs_python_k17.txt 2>/dev/null
echo "[6] microbenchmarks"; timeout -k 5 100 ./tools/stream_concurrency > $out/stream_concurrency.txt 2>&1; timeout -k 5 100 ./tools/ubench_mfma_price > $out/ubench_mfma_price.txt 2>&1; timeout -k 5 100 ./tools/pmc_calib > $out/pmc_calib.txt 2>&1
timeout -k 5 100 ./tools/ubench_chain > $out/ubench_chain.txt 2>&1; timeout -k 10 200 python tools/ntt_bench.py > $out/ntt_bench.txt 2>/dev/null
echo "[6b] clock ramp: the step bench by untimed work before the timed region"
for p in 0 0.3 1.0 3.0; do timeout -k 10 200 python bench.py --no-cpu-baseline --proof-k 0 --proofs 0 --preheat-s $p 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('preheat %.1f s (%d steps): %.1f Mpoints/s, %.4f ms per step; k_msm_accum0 alone %.4f ms' % (d['preheat']['seconds'], d['preheat']['steps'], d['value'], d['ms_per_step'], d['roofline']['avg_kernel_ms']))
"; done > $out/clock_ramp.txt
echo "[7] two ranks on one GPU (gloo)"
timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --dist-backend gloo --force-device 0 --proofs 16 --no-cpu-baseline > $out/bench_2rank_one_gpu_gloo.log 2>&1 || echo "2-rank run failed"
rm -f $out/kp/kp_kernel_trace.csv $out/kt/kt_kernel_trace.csv $out/k1/k1_kernel_trace.csv $out/bt/bt_kernel_trace.csv; rm -rf $out/*/*agent_info*
ls -la $out
