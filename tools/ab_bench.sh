#!/bin/bash
# A/B runs of bench.py on ONE box: tools/ab_bench.sh OUTDIR "ENV1=.. ENV2=.." "ENV=.." ...   ("-" = no extra environment)
# each configuration runs REPS times (default 2), interleaved; MSM leg only unless AB_ARGS says otherwise
out=$1; shift
mkdir -p "$out"
reps=${REPS:-2}
args=${AB_ARGS:---steps 20 --warmup 5 --proofs 0 --no-cpu-baseline}
for r in $(seq 1 $reps); do
  i=0
  for cfg in "$@"; do
    i=$((i+1))
    if [ "$cfg" = "-" ]; then e=""; else e="$cfg"; fi
    env $e python3 bench.py $args > "$out/cfg${i}_rep${r}.json" 2> "$out/cfg${i}_rep${r}.err"
    python3 - "$out/cfg${i}_rep${r}.json" "$cfg" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    p=d.get("proof") or {}
    print("%-50s %7.1f Mpts/s %6.3f ms/step acc live %.3f iso %.3f stages %s proof p50 %s pps %s thr %s" % (sys.argv[2], d["value"]/1e6, d["ms_per_step"], d["roofline"]["kernel_ms"], d["roofline"]["kernel_ms_isolated"], " ".join("%s=%.3f" % (k[4:],v) for k,v in d["stage_ms_isolated"].items()), p.get("p50_ms"), p.get("proofs_per_s"), (p.get("throughput_mode") or {}).get("proofs_per_s")))
except Exception as e:
    print("%-50s FAILED %r" % (sys.argv[2], e))
PY
  done
done
