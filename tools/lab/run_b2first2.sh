#!/bin/bash
for r in 1 2 3 4 5 6; do
  for e in "" "K16_B2_FIRST=1"; do
    env $e python3 tools/bench_proof.py --proofs 60 --no-stats 2>/dev/null | python3 -c "
import json,sys
l=[x for x in sys.stdin.read().splitlines() if x.startswith('{')][0]; d=json.loads(l); print('%-16s p50 %.3f ms  p99 %.2f  %.1f proofs/s' % ('$e' or '-', d['p50_ms'], d['p99_ms'], d['value']))"
  done
done
