#!/bin/bash
# previous build (tools/ab/libdehalo_prev.so) against the tree's on the host-witness paths
for lib in prev new; do
  if [ $lib = prev ]; then export DEHALO_LIBRARY=$PWD/tools/ab/libdehalo_prev.so; else unset DEHALO_LIBRARY; fi
  echo "== $lib"; python3 tools/host_advice_bench.py 17 30 2>/dev/null
done
