#!/bin/bash
O=gpurun_out/r05; mkdir -p $O
python3 -m pytest tests/test_gpu_parity.py -q -m gpu -k "compact_witness or keyless_shape" 2>&1 | tail -2
python3 tools/prove_soak.py 2000 2 > $O/prove_soak_keyless_shape_final.log 2>&1; tail -2 $O/prove_soak_keyless_shape_final.log
python3 tools/prove_fuzz.py 300 > $O/prove_fuzz_300_final.json 2>&1; tail -c 300 $O/prove_fuzz_300_final.json
