#!/bin/bash
# A/B of NTT tile shapes on the GPU box: rebuilds the four ntt units with the given -D flags and runs tools/ntt_bench.py each time
set -e
echo "== baseline"; python3 tools/ntt_bench.py
for v in "$@"; do
  echo "== $v"
  rm -f delay-encryption-in-halo2_amd/csrc/obj/ntt_*.o
  make -j8 delay-encryption-in-halo2_amd/libdehalo.so HIPFLAGS="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -Wall -Wno-unused-function -Werror=unused-variable -ffp-contract=off $v" > gpurun_out/ab_build.log 2>&1
  python3 tools/ntt_bench.py
done
