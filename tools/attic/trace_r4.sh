#!/bin/bash
# kernel traces of one k = 17 proof and of the MSM + NTT step (4 in flight) for a given environment; tools/trace_r4.sh TAG [ENV=VALUE ...]
tag=$1; shift
for kv in "$@"; do export "$kv"; done
out=gpurun_out/trace_$tag; rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kp -o kp -- python3 tools/profile_native_proof.py 17 delay_enc 20 > $out/proof.log 2>&1
f=$(find $out/kp -name "kp_kernel_trace.csv" | head -1)
python3 tools/timeline.py $f > $out/proof_timeline.txt
s=$(find $out/kp -name "kp_kernel_stats.csv" | head -1); cp $s $out/proof_kernel_stats.csv
grep "k = 17" $out/proof.log
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $out/ks -o ks -- python3 bench.py --in-process --no-single-stream --no-cpu-baseline --proof-k 0 --proofs 0 --preheat-s 0.3 > $out/step.log 2>&1
f=$(find $out/ks -name "ks_kernel_trace.csv" | head -1)
python3 tools/step_concurrency.py $f > $out/step_concurrency.txt
cat $out/step_concurrency.txt
python3 - $f > $out/step_window.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
acc = [i for i, r in enumerate(rows) if 'k_msm_accum0' in r['Kernel_Name']]
a, b = acc[-6], acc[-2]
t0 = int(rows[a]['Start_Timestamp'])
for r in rows[a - 5:b + 12]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print('%9.1f %8.1f  %-40s q%s grid %s' % ((s - t0) / 1e3, (e - s) / 1e3, r['Kernel_Name'].split('(')[0][-40:], r.get('Queue_Id'), r['Grid_Size_X']))
PY
rm -rf $out/kp $out/ks
