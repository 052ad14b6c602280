#!/bin/bash
python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_classes.py -q -m gpu -k "keyless_shape or prove or proof" 2>&1 | tail -2
python3 tools/prove_fuzz.py 200 31 2>&1 | tail -c 200; echo
for r in 1 2 3 4 5; do
  for e in "K16_H_WAIT_FIRST=1" ""; do
    env $e python3 tools/bench_proof.py --proofs 60 --no-stats 2>/dev/null | python3 -c "
import json,sys
l=[x for x in sys.stdin.read().splitlines() if x.startswith('{')][0]; d=json.loads(l); print('%-20s p50 %.3f ms  p99 %.2f  %.1f proofs/s' % ('$e' or '-', d['p50_ms'], d['p99_ms'], d['value']))"
  done
done
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf /tmp/k16_tl; rocprofv3 --kernel-trace --output-format csv -d /tmp/k16_tl -- python3 tools/bench_proof.py --proofs 6 > /dev/null 2>&1
python3 tools/proof_timeline.py /tmp/k16_tl 2 | grep -B3 -A3 "k_hscalars" | head -12
