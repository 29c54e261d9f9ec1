#!/bin/bash
# per-class times of k_msm_merge_all (DEHALO_MSM_MERGE_SPLIT=1: one launch per class, in the order block, wave, 32 lanes, light) for one MSM shape (tools/msm_small.py arguments)
. tools/exp_lib.sh      # the switches below exist in the measurement build only (make EXPERIMENTS=1)
export TMPDIR=/tmp DEHALO_MSM_MERGE_SPLIT=1
rm -rf gpurun_out/abk; rocprofv3 --kernel-trace --output-format csv -d gpurun_out/abk -o abk -- python3 tools/msm_small.py "$@" > /dev/null 2>&1
python3 - <<PY
import csv,glob
f=glob.glob('gpurun_out/abk/**/abk_kernel_trace.csv', recursive=True)[0]
rows=sorted(csv.DictReader(open(f)), key=lambda r:int(r['Start_Timestamp']))
m=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3 for r in rows if 'k_msm_merge_all' in r['Kernel_Name']]
m=m[-40:]
names=['block','wave','32 lanes','light']
print("$@:", ", ".join("%s %.1f us" % (names[i], sum(m[i::4])/len(m[i::4])) for i in range(4)))
PY
rm -rf gpurun_out/abk
