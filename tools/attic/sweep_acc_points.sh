#!/bin/bash
# points-per-lane rule of the accumulation (DEHALO_MSM_ACC_POINTS; 0 = rounds of msm_acc_waves waves): headline line and the k=17 proof
. tools/exp_lib.sh      # the switches below exist in the measurement build only (make EXPERIMENTS=1)
need_switch DEHALO_MSM_ACC_POINTS 
for p in 0 36 44 48 56 64; do
  export DEHALO_MSM_ACC_POINTS=$p
  echo "== acc_points $p"
  timeout -k 10 300 python bench.py --no-cpu-baseline --proof-k 0 --proofs 0 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('value',d['value'],'ms_per_step',d['ms_per_step'],'single',d.get('single_stream',{}).get('ms_per_step'),d.get('single_stream',{}).get('kernel_ms'))"
  timeout -k 10 300 python tools/profile_proof.py 17 1 10 2>/dev/null | grep "side context" | cut -c1-200
done
