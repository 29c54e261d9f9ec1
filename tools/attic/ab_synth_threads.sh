#!/bin/bash
for t in 1 4 8; do DEHALO_SYNTH_TRACE=1 DEHALO_SYNTH_THREADS=$t python3 tools/synth_bench.py 2>&1 | tail -7; done
