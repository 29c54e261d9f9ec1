#!/bin/bash
# Round 5's tree (tools/ab/r05tree: `git archive 9e42dcc`, built) against this one on ONE box, alternating: the step, and the k = 17 / k = 20 delay_enc proofs.
#   bash tools/ab_r05_r06.sh [rounds] > gpurun_out/ab_r05_r06.txt
rounds=${1:-3}
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read())
p=d.get('proof') or {}
o={q['k']: q['gpu_ms'] for q in (d.get('proof_other_k') or [])}
print('$1: %.1f Mpoints/s, %.4f ms/step; alone sort %.4f acc %.4f red %.4f ntt %.4f, one step %.4f; k17 proof %s ms (%s rows), k20 %s, k14 %s; batch %s proofs/s' % (d['value'], d['ms_per_step'], d['alone_ms']['msm_sort'], d['alone_ms']['msm_accumulate'], d['alone_ms']['msm_reduce'], d['alone_ms']['ntt'], d['single_stream_ms_per_step'], p.get('gpu_ms'), p.get('rows'), o.get(20), o.get(14), (d.get('batch_proofs') or {}).get('proofs_per_s')))"; }
for r in $(seq 1 $rounds); do
  (cd tools/ab/r05tree && python bench.py --no-cpu-baseline --no-verify --full-out "" 2>/dev/null) | line "round $r  round-5 build"
  python bench.py --no-cpu-baseline --no-verify --full-out "" 2>/dev/null | line "round $r  round-6 build"
done
