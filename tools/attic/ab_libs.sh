#!/bin/bash
# A/B of whole-library builds on the GPU box, alternating so that box / clock state cancels: tools/ab_libs.sh ROUNDS "CMD" name1="flags" name2="flags" ...
# Every variant is built once (all units, its own object directory) into gpurun_out/ab/<name>/libdehalo.so; then ROUNDS times, in turn, CMD runs with
# DEHALO_LIBRARY pointing at each.  Example: tools/ab_libs.sh 3 "python3 tools/ntt_bench.py" base="" serial="-DNTT_X_SERIAL_LOADS"
set -e
rounds=$1; cmd=$2; shift 2
BASE="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -Wall -Wno-unused-function -ffp-contract=off"
names=()
for spec in "$@"; do
  name=${spec%%=*}; flags=${spec#*=}
  names+=("$name")
  mkdir -p gpurun_out/ab/$name
  make -j16 gpurun_out/ab/$name/libdehalo.so LIB=gpurun_out/ab/$name/libdehalo.so OBJDIR=gpurun_out/ab/$name/obj HIPFLAGS="$BASE $flags" > gpurun_out/ab/$name/build.log 2>&1 || { tail -20 gpurun_out/ab/$name/build.log; exit 1; }
  echo "built $name ($flags)"
done
for r in $(seq 1 $rounds); do
  for name in "${names[@]}"; do
    echo "== $name, round $r"
    DEHALO_LIBRARY=$PWD/gpurun_out/ab/$name/libdehalo.so $cmd 2>&1 | grep -v amdgpu.ids
  done
done
