#!/bin/bash
# Round 6, VERDICT r5 item 3: the H MSM's place in a proof's schedule, re-measured in the regime the product runs in (eight
# hardware queues: tools/bench_proof.py calls k16_runtime_hw_queues(8)) with the switches IN THE TREE (csrc/ctx.hip from_env):
#   K16_H_LANE=3        the H MSM on a lane (stream + workspace) of its own instead of behind C's MSM on lane 1
#   K16_H_WAIT_FIRST=1  the H MSM's wait for the chain issued behind its sort's memset instead of in front of it
#   K16_B1_LANE=3       B1's MSM on a lane of its own instead of behind A's on lane 0 (its tail then runs under the chain)
# Five alternating runs of 60 proofs each + two provers sharing the GPU + one kernel timeline per variant.
out=${1:-gpurun_out/r6_hlane}
mkdir -p "$out"
variants=("" "K16_H_LANE=3" "K16_H_WAIT_FIRST=1" "K16_B1_LANE=3" "K16_B1_LANE=3 K16_H_WAIT_FIRST=1")
for v in "${variants[@]}"; do
  env $v python3 -m pytest tests/test_gpu_parity.py -q -m gpu -k "keyless_shape_proof_full_size" 2>&1 | tail -1 | sed "s/^/[parity ${v:-default}] /"
done | tee "$out/parity.log"
for r in 1 2 3 4 5; do
  for v in "${variants[@]}"; do
    env $v python3 tools/bench_proof.py --proofs 60 --no-stats --random-rs 2>/dev/null | python3 -c "
import json,sys
l=[x for x in sys.stdin.read().splitlines() if x.startswith('{')][0]; d=json.loads(l); print('%-36s p50 %.3f ms  p99 %.2f  %.1f proofs/s' % ('${v:-default}', d['p50_ms'], d['p99_ms'], d['value']))"
  done
done | tee "$out/ab_latency.log"
for v in "${variants[@]}"; do
  env $v python3 tools/bench_proof.py --proofs 60 --concurrent 2 --no-stats 2>/dev/null | grep "throughput mode" | cut -c1-220 | sed "s/^/[two provers ${v:-default}] /"
done | tee "$out/ab_two_provers.log"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
i=0
for v in "${variants[@]}"; do
  rm -rf /tmp/k16_tl; env $v rocprofv3 --kernel-trace --output-format csv -d /tmp/k16_tl -- python3 tools/bench_proof.py --proofs 6 --no-stats > /dev/null 2>&1
  { echo "variant: ${v:-default}"; python3 tools/proof_timeline.py /tmp/k16_tl 3; } > "$out/proof_timeline_$i.txt"
  i=$((i+1))
done
