#!/bin/bash
# Shader clock and package power while proofs run (read-only rocm-smi queries from an ordinary user):
#   tools/lab/clocks_under_load.sh [provers]      -> gpurun_out/clocks_<provers>.log
# The proof-level issue bound in DESIGN.md 6.000 is quoted at 2.4 GHz; this says what the clock really is under that load.
conc=${1:-2}
out=gpurun_out/clocks_$conc.log
mkdir -p gpurun_out
extra=""; [ "$conc" -gt 1 ] && extra="--concurrent $conc"
python3 tools/bench_proof.py --proofs 1500 --no-stats $extra > gpurun_out/clocks_bench_$conc.json 2>/dev/null &
pid=$!
: > $out
SECONDS=0
# one leg with a single prover (1500 proofs, ~9 s) is followed by the leg with $conc provers; every sample carries its time
while kill -0 $pid 2>/dev/null; do
  printf "t=%3d s  " $SECONDS >> $out
  rocm-smi -c -P 2>/dev/null | grep -E "sclk|mclk|ower" | sed 's/=//g; s/  */ /g' | tr '\n' ' ' >> $out
  echo >> $out
  sleep 1
done
wait $pid
grep -h '^{' gpurun_out/clocks_bench_$conc.json | python3 -c '
import json,sys
for l in sys.stdin:
    o=json.loads(l); print(o["metric"], ": %.1f proofs/s, p50 %.2f ms" % (o["value"], o["p50_ms"]))' >> $out
cat $out
