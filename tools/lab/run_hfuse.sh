#!/bin/bash
# Round 5 experiment (DESIGN_HISTORY.md): the H scalars formed by the chain's last pass.  Needs a tree with
# tools/lab/ntt_h_epilogue.patch applied -- K16_NTT_H_UNFUSED is that patch's switch (it selects the product's unfused path), not one of csrc/.
python3 -m pytest tests/test_gpu_parity.py tests/test_boundary.py -q -m gpu -k "ntt or keyless_shape or prove or proof or toy" 2>&1 | tail -2
python3 tools/prove_fuzz.py 300 11 2>&1 | tail -c 220; echo
REPS=${REPS:-4} tools/lab/run_ab_tailfit.sh K16_NTT_H_UNFUSED=1 - 2>&1 | sed -e "s/stages.*proof/proof/" | cut -c1-22,50-200
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf /tmp/k16_tl; rocprofv3 --kernel-trace --output-format csv -d /tmp/k16_tl -- python3 tools/bench_proof.py --proofs 6 > /dev/null 2>&1
python3 tools/proof_timeline.py /tmp/k16_tl 2 | grep "ntt_pass\|hscalars\|PartCfgFlat5\|k_mul\|spmv" | head -20
