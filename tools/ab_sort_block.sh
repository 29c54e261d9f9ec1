#!/bin/bash
# The step (2^20 Pallas MSM + 2^20 NTT, four in flight) with the sort's workgroups at 1024 / 512 threads (dehalo_ctx_set_tuning msm_sort_block), alternating runs on
# ONE box so that box-to-box differences (+-3 %) cancel: bash tools/ab_sort_block.sh [rounds] > gpurun_out/ab_sort_block.txt
rounds=${1:-4}
for r in $(seq 1 $rounds); do
  for sb in 1024 512; do
    python bench.py --proof-k 0 --no-cpu-baseline --sort-block $sb --full-out "" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('round $r sort_block $sb: %.1f Mpoints/s, %.4f ms/step; alone sort %.4f acc %.4f red %.4f ntt %.4f; overlapped sort %.3f acc %.3f red %.3f ntt %.3f' % (d['value'], d['ms_per_step'], d['alone_ms']['msm_sort'], d['alone_ms']['msm_accumulate'], d['alone_ms']['msm_reduce'], d['alone_ms']['ntt'], d['overlapped_ms']['msm_sort'], d['overlapped_ms']['msm_accumulate'], d['overlapped_ms']['msm_reduce'], d['overlapped_ms']['ntt']))"
  done
done
