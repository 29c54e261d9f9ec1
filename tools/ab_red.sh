#!/bin/bash
# bucket reduction: quads against one lane per 4-bucket block, by batch size.  NOTE: needs a build of msm.cuh that reads DEHALO_MSM_RED_QUAD_MAX (the lanes up to which the quad
# form runs) in place of its fixed threshold -- the experiment of profiles/r03_bucket_reduction_quads_vs_lanes.txt; the shipped library ignores the variable.
export TMPDIR=/tmp
for b in 4 6 7 8 10 12; do
for q in 0 1000000; do
  export DEHALO_MSM_RED_QUAD_MAX=$q
  rm -rf gpurun_out/abk; rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/abk -o abk -- python3 tools/msm_small.py 17 uniform $b > gpurun_out/abk.log 2>&1
  python3 - <<PY
import csv,glob
f=glob.glob('gpurun_out/abk/**/abk_kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'reduce_local' in r['Name']: print("batch $b quad_max $q: %-44s %9.1f us" % (r['Name'].split('(')[0][:44], float(r['AverageNs'])/1e3))
PY
  grep "min" gpurun_out/abk.log
done; done
rm -rf gpurun_out/abk
