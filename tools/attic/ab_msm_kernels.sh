#!/bin/bash
# per-kernel times of one MSM shape under rocprofv3 for the previous and the current build (tools/ab/libdehalo_prev.so)
export TMPDIR=/tmp
shape="$@"
for lib in prev new; do
  if [ $lib = prev ]; then export DEHALO_LIBRARY=$PWD/tools/ab/libdehalo_prev.so; else unset DEHALO_LIBRARY; fi
  rm -rf gpurun_out/abk; rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/abk -o abk -- python3 tools/msm_small.py $shape > /dev/null 2>&1
  echo "== $lib: $shape"
  python3 - <<PY
import csv,glob
f=glob.glob('gpurun_out/abk/**/abk_kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'msm' in r['Name'] and 'build_table' not in r['Name']: print("  %-48s %5s %9.1f us" % (r['Name'].split('(')[0][:48], r['Calls'], float(r['AverageNs'])/1e3))
PY
done
rm -rf gpurun_out/abk
