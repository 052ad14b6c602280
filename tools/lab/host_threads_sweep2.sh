#!/bin/bash
# pack + issue time of the witness hand-off and proof p50 against the width of the host pool, beyond the default's 12
for r in 1 2; do for t in 8 12 16 24 32; do
  K16_HOST_THREADS=$t K16_TRACE_HOST=1 python3 tools/bench_proof.py --proofs 16 2> /tmp/h.err | python3 -c "
import json,sys
l=[x for x in sys.stdin.read().splitlines() if x.startswith('{')][0]; d=json.loads(l); print('threads %2d: p50 %.3f ms  %.1f proofs/s' % ($t, d['p50_ms'], d['value']), end='')"
  grep "witness upload issued" /tmp/h.err | tail -10 | awk '{s+=$6} END {printf "   pack + issue %.0f us\n", s/NR}'
done; done
nproc; lscpu | grep -E "Model name|Socket|Core|Thread|NUMA node\(s\)|L3"
