#!/bin/bash
python3 -m pytest tests/test_gpu_classes.py -q -m gpu -k "b2_first" 2>&1 | tail -2
REPS=${REPS:-4} tools/lab/run_ab_tailfit.sh - K16_B2_FIRST=1 2>&1 | sed -e "s/stages.*proof/proof/" | cut -c1-16,50-200
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export K16_B2_FIRST=1
rm -rf /tmp/k16_tl; rocprofv3 --kernel-trace --output-format csv -d /tmp/k16_tl -- python3 tools/bench_proof.py --proofs 6 > /dev/null 2>&1
python3 tools/proof_timeline.py /tmp/k16_tl 2 | grep "accumulate\|Eng2n\|k_part_stage\|k_bins_stage\|hscalars" | head -30
