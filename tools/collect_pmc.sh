#!/bin/bash
# HBM traffic of k_msm_accum0: FETCH_SIZE and WRITE_SIZE in separate --pmc passes over the bench, the same two passes over
# tools/pmc_calib (known byte counts), then tools/pmc_summary.py writes profiles/pmc_traffic.json.  usage: tools/collect_pmc.sh KERNEL_REV [WINDOW_BITS of the 2^20 table the library chooses: 17]
set -e
rev=${1:?kernel revision (bench.py KERNEL_REV)}
wb=${2:-17}
out=gpurun_out/pmc; rm -rf $out; mkdir -p $out
export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/main_$c -- python3 bench.py --in-process --no-cpu-baseline --proof-k 0 --proofs 0 --steps 5 --warmup 2 --preheat-s 0 > $out/main_$c.log 2>&1
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/calib_$c -- ./tools/pmc_calib > $out/calib_$c.log 2>&1
done
mkdir -p $out/main $out/calib
for c in FETCH_SIZE WRITE_SIZE; do
  cp $(find $out/main_$c -name '*counter_collection.csv' | head -1) $out/main/${c}_counter_collection.csv
  cp $(find $out/calib_$c -name '*counter_collection.csv' | head -1) $out/calib/${c}_counter_collection.csv
done
python3 tools/pmc_summary.py $out/main --json $out/pmc_traffic.json --log-n 20 --curve pallas --window-bits $wb --kernel-rev $rev --calib $out/calib | tail -2
