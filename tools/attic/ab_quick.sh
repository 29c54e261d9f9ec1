#!/bin/bash
# the four numbers an MSM / prover change is judged by, two rounds: bench step, k = 17 delay_enc, K = 11 pose_enc, 64-proof batch.   tools/ab_quick.sh [ENV=VALUE ...]
for kv in "$@"; do export "$kv"; done
for round in 1 2; do
  timeout -k 10 200 python bench.py --in-process --no-cpu-baseline --proof-k 0 --proofs 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('step: %.1f Mpoints/s, %.4f ms per step; alone %.4f ms' % (d['value'], d['ms_per_step'], d['single_stream']['ms_per_step']), d.get('breakdown_ms_per_step'))"
  timeout -k 10 200 python tools/profile_native_proof.py 17 delay_enc 60 2>/dev/null | grep "k = 17"
  timeout -k 10 200 python tools/profile_native_proof.py 11 pose_enc 60 2>/dev/null | grep "k = 11"
  timeout -k 10 200 python tools/batch_trace.py 17 4 64 0 1 2>/dev/null | grep batch
done
