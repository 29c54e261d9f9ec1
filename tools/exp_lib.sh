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

# need_switch NAME ...: every DEHALO_* switch an A/B script sets must still be a string of the measurement library -- a switch deleted from the sources
# would make both arms of the A/B the same code and the script would print noise as if it were a result
need_switch() {
  for name in "$@"; do
    if ! strings "$DEHALO_LIBRARY" | grep -qx "$name"; then
      echo "exp_lib: the measurement library does not read $name (switch removed from the sources?)" >&2
      exit 2
    fi
  done
}
