#!/bin/bash
A=$GRAFT_REPO_ROOT/keyless-zk-proofs_amd/alt
python3 -m pytest tests/test_gpu_parity.py -q -m gpu -k "ntt or keyless_shape or prove or lazy" 2>&1 | tail -2
python3 tools/ntt_fuzz.py 600 21 2>&1 | tail -c 200; echo
python3 tools/prove_fuzz.py 300 21 2>&1 | tail -c 200; echo
for i in 1 2 3; do
  python3 tools/ntt_timing.py 21 30 | tail -1
  K16_LIB_PATH=$A/libk16_nttmont.so python3 tools/ntt_timing.py 21 30 | tail -1 | sed 's/^/   montgomery twiddles: /'
done
REPS=4 tools/lab/run_ab_tailfit.sh - K16_LIB_PATH=$A/libk16_nttmont.so 2>&1 | sed -e "s/stages.*proof/proof/" -e "s/K16_LIB_PATH=.*alt.libk16_//" | cut -c1-14,50-200
