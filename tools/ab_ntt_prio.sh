#!/bin/bash
# lone and batched transforms with workgroup issue priorities by slot (DEHALO_NTT_PRIO), and the phases of the workgroups (DEHALO_NTT_STAMPS)
. tools/exp_lib.sh      # the switches below exist in the measurement build only (make EXPERIMENTS=1)
for round in 1 2; do
for m in 0 1 2; do echo "== DEHALO_NTT_PRIO=$m, round $round"; DEHALO_NTT_PRIO=$m timeout -k 10 200 python tools/ntt_phases.py 2>/dev/null | grep "min"; done
done
for m in 0 1 2; do echo "== phases, DEHALO_NTT_PRIO=$m"; DEHALO_NTT_PRIO=$m DEHALO_NTT_STAMPS=1 timeout -k 10 200 python tools/ntt_phases.py pasta_fp 20 1 2>&1 | grep "ntt pass"; done
echo "== phases of the batch 23 x 2^19"; DEHALO_NTT_STAMPS=1 timeout -k 10 200 python tools/ntt_phases.py bn254_fr 19 23 2>&1 | grep "ntt pass"
