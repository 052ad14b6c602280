#!/bin/bash
K16_H_LANE=3 python3 -m pytest tests/test_gpu_parity.py -q -m gpu -k "keyless_shape" 2>&1 | tail -1
for r in 1 2 3 4 5; do
  for e in "" "K16_H_LANE=3"; do
    env $e python3 tools/bench_proof.py --proofs 60 --no-stats 2>/dev/null | python3 -c "
import json,sys
l=[x for x in sys.stdin.read().splitlines() if x.startswith('{')][0]; d=json.loads(l); print('%-14s p50 %.3f ms  p99 %.2f  %.1f proofs/s' % ('$e' or '-', d['p50_ms'], d['p99_ms'], d['value']))"
  done
done
for e in "" "K16_H_LANE=3"; do env $e python3 tools/bench_proof.py --proofs 60 --concurrent 2 --no-stats 2>/dev/null | tail -1 | cut -c1-160; done
