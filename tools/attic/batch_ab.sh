for p in 0 48 36; do
. tools/exp_lib.sh      # the switches below exist in the measurement build only (make EXPERIMENTS=1)
need_switch DEHALO_MSM_ACC_POINTS 
  export DEHALO_MSM_ACC_POINTS=$p
  for rep in 1 2; do
  timeout -k 10 400 python bench.py --no-cpu-baseline --no-single-stream --no-verify --proofs 16 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('acc_points $p', 'batch', d['batch_proofs']['proofs_per_s'], 'proof', d['proof']['gpu_ms'], d['proof']['gpu_ms_median'])"
  done
done
