#!/bin/bash
# samples rocm-smi power / clocks while a bench configuration runs:  tools/lab/power_probe.sh "ENV=.. ENV=.." [steps]
cfg="$1"; steps=${2:-4000}
if [ "$cfg" = "-" ]; then cfg=""; fi
env $cfg K16_BENCH_NOCHECK=1 python3 bench.py --steps $steps --warmup 5 --proofs 0 --no-cpu-baseline > /tmp/pp.json 2>/tmp/pp.err &
pid=$!
sleep 16
for i in 1 2 3 4 5 6 7 8; do
  rocm-smi --showpower --showclocks --json 2>/dev/null | python3 -c '
import json,sys
try:
    d=json.load(sys.stdin); c=d[sorted(d)[0]]
    print("   ", {k:v for k,v in c.items() if "ower" in k or "sclk" in k or "mclk" in k or "fclk" in k})
except Exception as e: print("    smi parse failed", e)'
  sleep 1.0
done
wait $pid
python3 -c '
import json
d=json.loads(open("/tmp/pp.json").read().strip().splitlines()[-1])
print("    ->", "%.1f Mpts/s %.3f ms/step acc live %.3f" % (d["value"]/1e6, d["ms_per_step"], d["roofline"]["kernel_ms"]))'
