#!/bin/bash
# Round 6: the witness MSMs' G2 accumulation with two / four lane pairs per segment (K16_G2_ACC_SPLIT, msm_kernels.inc
# k_accumulate_split) against one -- B2's chain is what the H MSM's start waits for (profiles/r06/proof_timeline_8queues_*).
out=${1:-gpurun_out/r6_g2split}
mkdir -p "$out"
variants=("" "K16_G2_ACC_SPLIT=2" "K16_G2_ACC_SPLIT=4" "K16_G2_ACC_SPLIT=2 K16_B1_LANE=3" "K16_G2_ACC_SPLIT=4 K16_B1_LANE=3")
for r in 1 2 3 4 5; do
  for v in "${variants[@]}"; do
    env $v python3 tools/bench_proof.py --proofs 60 --no-stats --random-rs --concurrent 2 2>/dev/null | python3 -c "
import json,sys
l=[json.loads(x) for x in sys.stdin.read().splitlines() if x.startswith('{')]
print('%-36s p50 %.3f ms  p99 %.2f  %.1f proofs/s | two provers %.1f proofs/s' % ('${v:-default}', l[0]['p50_ms'], l[0]['p99_ms'], l[0]['value'], l[1]['value']))"
  done
done | tee "$out/ab_latency.log"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
i=0
for v in "${variants[@]}"; do
  rm -rf /tmp/k16_tl; env $v rocprofv3 --kernel-trace --output-format csv -d /tmp/k16_tl -- python3 tools/bench_proof.py --proofs 6 --no-stats > /dev/null 2>&1
  { echo "variant: ${v:-default}"; python3 tools/proof_timeline.py /tmp/k16_tl 3; } > "$out/proof_timeline_$i.txt"
  i=$((i+1))
done
