#!/bin/bash
# A/B of compile-time variants on the GPU box: for every argument (a string of -D flags, "" = as committed) rebuild the library
# and print the headline line's value / ms_per_step / single-stream kernel times.   usage: tools/ab_flags.sh "" "-DX=1" ...
set -e
BASE="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -Wall -Wno-unused-function -ffp-contract=off"
for v in "$@"; do
  echo "== [$v]"
  rm -f delay-encryption-in-halo2_amd/csrc/obj/msm_*.o delay-encryption-in-halo2_amd/csrc/obj/ntt_*.o
  make -j16 delay-encryption-in-halo2_amd/libdehalo.so HIPFLAGS="$BASE $v" > gpurun_out/ab_build.log 2>&1 || { tail -5 gpurun_out/ab_build.log; exit 1; }
  for rep in 1 2; do
  timeout -k 10 300 python bench.py --no-cpu-baseline --proof-k 0 --proofs 0 ${BENCH_ARGS} 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('value',d['value'],'ms_per_step',d['ms_per_step'],'single',d.get('single_stream',{}).get('ms_per_step'),d.get('single_stream',{}).get('kernel_ms'))"
  done
done
