#!/bin/bash
# A/B runs of the Keyless-shape proof on ONE box: tools/ab_proof.sh "ENV1=.. ENV2=.." "ENV=.." ...   ("-" = no extra environment)
# REPS interleaved repetitions (default 3); AB_CONC=4 adds the throughput mode (that many provers sharing the GPU)
reps=${REPS:-3}
conc=${AB_CONC:-0}
proofs=${AB_PROOFS:-30}
for r in $(seq 1 $reps); do
  for cfg in "$@"; do
    if [ "$cfg" = "-" ]; then e=""; else e="$cfg"; fi
    extra=""; [ "$conc" -gt 1 ] && extra="--concurrent $conc"
    env $e python3 tools/bench_proof.py --proofs $proofs --no-stats $extra 2>/dev/null | python3 -c '
import json,sys
o=[json.loads(l) for l in sys.stdin if l.startswith("{")]
s="%-60s p50 %.3f ms  %.1f proofs/s" % (sys.argv[1], o[0]["p50_ms"], o[0]["value"])
if len(o)>1: s+="   throughput mode %.1f proofs/s p50 %.2f" % (o[1]["value"], o[1]["p50_ms"])
print(s)' "$cfg"
  done
done
