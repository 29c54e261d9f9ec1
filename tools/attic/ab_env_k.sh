#!/bin/bash
# tools/ab_env_k.sh "K1 K2 ..." NAME VALUE_A VALUE_B ...: delay_enc-shaped proofs at the given k under an environment variable, three rounds
. tools/exp_lib.sh      # the switches below exist in the measurement build only (make EXPERIMENTS=1)
ks=$1; name=$2; shift 2
for round in 1 2 3; do
for v in "$@"; do
  if [ "$v" = unset ]; then unset $name; else export $name=$v; fi
  for k in $ks; do echo -n "$name=$v round $round: "; python3 tools/profile_native_proof.py $k delay_enc 60 2>/dev/null | grep "k = $k"; done
done; done
