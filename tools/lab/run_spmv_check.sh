#!/bin/bash
python3 -m pytest tests/test_gpu_parity.py -q -m gpu -k "keyless_shape or spmv or prove" 2>&1 | tail -2
REPS=${REPS:-3} tools/lab/run_ab_tailfit.sh - 2>&1 | sed -e "s/stages.*proof/proof/" | cut -c1-10,50-200
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf /tmp/k16_tl; rocprofv3 --kernel-trace --output-format csv -d /tmp/k16_tl -- python3 tools/bench_proof.py --proofs 6 > /dev/null 2>&1
python3 tools/proof_timeline.py /tmp/k16_tl 2 | head -12
