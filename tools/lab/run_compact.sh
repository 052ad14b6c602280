#!/bin/bash
python3 -m pytest tests/test_gpu_parity.py -q -m gpu -k "compact" 2>&1 | tail -3
K16_BENCH_NO_COLD=1 K16_BENCH_NO_HOST_LEG=1 python3 bench.py --steps 5 --warmup 2 --proofs 30 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d['proof']
print('p50 prove_mem %.3f ms | compact hand-off %s' % (p['p50_ms'], p.get('compact_hand_off')))"
K16_BENCH_NO_COLD=1 K16_BENCH_NO_HOST_LEG=1 python3 bench.py --steps 5 --warmup 2 --proofs 30 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d['proof']
print('p50 prove_mem %.3f ms | compact hand-off %s' % (p['p50_ms'], p.get('compact_hand_off')))"
