#!/bin/bash
# dehalo_create_proof_circuit with the upload streamed behind the witness writer (default) against synthesize-then-upload (DEHALO_SYNTH_STREAM=0)
python -m pytest tests/test_native.py -m gpu -x -q -k "end_to_end" 2>&1 | tail -1 || exit 1
for v in 0 1 0 1; do DEHALO_SYNTH_STREAM=$v python3 tools/circuit_call_bench.py 40 2>/dev/null | tail -4; done
