#!/bin/bash
A=$GRAFT_REPO_ROOT/keyless-zk-proofs_amd/alt
REPS=${REPS:-4} tools/lab/run_ab_tailfit.sh - K16_LIB_PATH=$A/libk16_${1}.so 2>&1 | sed -e "s/stages.*proof/proof/" -e "s/K16_LIB_PATH=.*alt.libk16_//" | cut -c1-16,50-200
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export K16_LIB_PATH=$A/libk16_${1}.so
rm -rf /tmp/k16_tl; rocprofv3 --kernel-trace --output-format csv -d /tmp/k16_tl -- python3 tools/bench_proof.py --proofs 6 > /dev/null 2>&1
python3 tools/proof_timeline.py /tmp/k16_tl 2 | grep "accumulate\|Eng2n\|k_part_stage\|k_bins_stage\|hscalars" | head -30
