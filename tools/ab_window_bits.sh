#!/bin/bash
# The step (2^20 Pallas MSM + 2^20 NTT, four in flight) by the window of the resident table (dehalo_bases_register window_bits; 0 = the library's choice),
# alternating runs on ONE box:  bash tools/ab_window_bits.sh "16 17" 4 > gpurun_out/ab_window_bits.txt
windows=${1:-"16 17"}
rounds=${2:-3}
for r in $(seq 1 $rounds); do
for wb in $windows; do
python bench.py --proof-k 0 --no-cpu-baseline --window-bits $wb --full-out "" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('round $r window $wb: %.1f Mpoints/s, %.4f ms/step; alone sort %.4f acc %.4f red %.4f ntt %.4f single %.4f' % (d['value'], d['ms_per_step'], d['alone_ms']['msm_sort'], d['alone_ms']['msm_accumulate'], d['alone_ms']['msm_reduce'], d['alone_ms']['ntt'], d['single_stream_ms_per_step']))"
done; done
