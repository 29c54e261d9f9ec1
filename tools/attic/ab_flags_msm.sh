#!/bin/bash
# A/B of compile-time variants of the MSM units on the GPU box: rebuilds msm_bn254 with the given -D flags and runs tools/tails_by_batch.sh each time
set -e
SHAPES=${SHAPES:-"1 3 4 7 10"}
for v in "$@"; do
  echo "== $v"
  rm -f delay-encryption-in-halo2_amd/csrc/obj/msm_bn254.o
  make -j8 delay-encryption-in-halo2_amd/libdehalo.so HIPFLAGS="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -Wall -Wno-unused-function -ffp-contract=off $v" > gpurun_out/ab_build.log 2>&1
  bash tools/tails_by_batch.sh $SHAPES
done
