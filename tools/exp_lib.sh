#!/bin/bash
# Sourced by the A/B scripts that switch behaviour through DEHALO_* environment variables: since round 5 the shipped library reads none of them (csrc/internal.hpp),
# so these scripts run against the MEASUREMENT build -- make EXPERIMENTS=1 into gpurun_out/ab/exp/, built here on first use -- through DEHALO_LIBRARY.
#   . tools/exp_lib.sh        (no-op if DEHALO_LIBRARY is set already)
if [ -z "$DEHALO_LIBRARY" ]; then
  mkdir -p gpurun_out/ab/exp
  if [ ! -f gpurun_out/ab/exp/libdehalo.so ] || [ -n "$(find delay-encryption-in-halo2_amd/csrc include -newer gpurun_out/ab/exp/libdehalo.so -type f 2>/dev/null | head -1)" ]; then
    make -j16 gpurun_out/ab/exp/libdehalo.so LIB=gpurun_out/ab/exp/libdehalo.so OBJDIR=gpurun_out/ab/exp/obj EXPERIMENTS=1 > gpurun_out/ab/exp/build.log 2>&1 || { tail -20 gpurun_out/ab/exp/build.log; exit 1; }
  fi
  export DEHALO_LIBRARY=$PWD/gpurun_out/ab/exp/libdehalo.so
  echo "# measurement build: $DEHALO_LIBRARY"
fi
