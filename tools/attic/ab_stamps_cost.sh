#!/bin/bash
# what the phase-stamp hooks in k_msm_bred / k_msm_merge2 cost when they are off at run time: the library as built, then the MSM units rebuilt with -DDEHALO_PHASE_STAMPS=0
echo "== hooks compiled in (off at run time)"; tools/ab_quick.sh
F="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -Wall -Wno-unused-function -ffp-contract=off -DDEHALO_PHASE_STAMPS=0"
touch delay-encryption-in-halo2_amd/csrc/msm_bred.cuh
make -j16 HIPFLAGS="$F" delay-encryption-in-halo2_amd/libdehalo.so > gpurun_out/rebuild.log 2>&1 || { echo "rebuild failed"; tail -5 gpurun_out/rebuild.log; exit 1; }
echo "== hooks compiled out"; tools/ab_quick.sh
