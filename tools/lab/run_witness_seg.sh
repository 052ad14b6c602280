#!/bin/bash
# Round 6: segment length of the witness MSMs' accumulations (K16_WITNESS_SEG; 32 since round 1) -- the B2 (G2) chain, 32
# dependent G2 additions per lane pair under the NTT passes, is what the H MSM's start waits for (profiles/r06/proof_timeline_*).
out=${1:-gpurun_out/r6_wseg}
mkdir -p "$out"
variants=("" "K16_WITNESS_SEG=16" "K16_WITNESS_SEG=24" "K16_WITNESS_SEG=48" "K16_WITNESS_SEG=16 K16_B1_LANE=3" "K16_B1_LANE=3")
for v in "${variants[@]}"; do
  env $v python3 -m pytest tests/test_gpu_parity.py -q -m gpu -k "keyless_shape_proof_full_size" 2>&1 | tail -1 | sed "s/^/[parity ${v:-default}] /"
done | tee "$out/parity.log"
for r in 1 2 3 4; do
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
