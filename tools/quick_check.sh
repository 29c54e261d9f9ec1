#!/bin/bash
# MSM parity subset, the single-step kernel timeline, the headline line and the k=17 proof time (GPU box)
set -e
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "msm or multiexp or commit or proof or prover" > gpurun_out/qc_test.log 2>&1 || { tail -20 gpurun_out/qc_test.log; exit 1; }
tail -1 gpurun_out/qc_test.log
rm -rf gpurun_out/stl
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/stl -o k -- python3 bench.py --in-process --inflight 1 --no-single-stream --no-cpu-baseline --proof-k 0 --proofs 0 --steps 5 --warmup 2 > gpurun_out/stl.log 2>&1
python3 tools/step_timeline.py gpurun_out/stl/k_kernel_trace.csv
timeout -k 10 300 python bench.py --no-cpu-baseline --proof-k 0 --proofs 0 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('value',d['value'],'ms_per_step',d['ms_per_step'],'single',d.get('single_stream',{}).get('ms_per_step'),d.get('single_stream',{}).get('kernel_ms'))"
timeout -k 10 300 python tools/profile_proof.py 17 1 10 2>/dev/null | tail -2 | cut -c1-330
