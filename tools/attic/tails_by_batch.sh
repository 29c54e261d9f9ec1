#!/bin/bash
# the MSM's kernels by shape: arguments are <columns> (2^17 points, uniform scalars) or <log n>:<distribution>:<columns> (tools/msm_small.py)
export TMPDIR=/tmp
for a in "$@"; do
  case $a in *:*) IFS=: read ln dist b <<< "$a";; *) ln=17; dist=uniform; b=$a;; esac
  rm -rf gpurun_out/abk; rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/abk -o abk -- python3 tools/msm_small.py $ln $dist $b > gpurun_out/abk.log 2>&1
  python3 - <<PY
import csv,glob
f=glob.glob('gpurun_out/abk/**/abk_kernel_stats.csv', recursive=True)[0]
d={r['Name'].split('(')[0].replace('void ','')[:30]: float(r['AverageNs'])/1e3 for r in csv.DictReader(open(f))}
order=['accum0','merge_classify','merge_all','reduce_local','tree_sum','hist','part','bucket']
pick=lambda n: next((v for k,v in d.items() if 'k_msm_'+n in k), 0.0)
print("2^%s %-8s x %2s: " % ("$ln", "$dist", "$b") + "  ".join("%s %.1f" % (n, pick(n)) for n in order) + "   | " + open('gpurun_out/abk.log').read().strip().split('\n')[-1].split(': ')[-1])
PY
done
rm -rf gpurun_out/abk
