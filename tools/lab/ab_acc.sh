#!/bin/bash
# round 5: A/B of the bucket accumulation's addition on ONE box -- new formulas (libk16.so), round 4's (alt/libk16_old.so),
# new with one accumulator chain per product column (alt/libk16_chain1.so).  Usage: tools/lab/ab_acc.sh OUTDIR
out=${1:-gpurun_out/ab_acc}; mkdir -p $out
A=keyless-zk-proofs_amd/alt
cfgs=("-")
for v in "$@"; do :; done
for name in old chain1 ${AB_EXTRA}; do [ -f $A/libk16_$name.so ] && cfgs+=("K16_LIB_PATH=$PWD/$A/libk16_$name.so"); done
REPS=${REPS:-2} tools/ab_bench.sh $out/bench "${cfgs[@]}" 2>&1 | tee $out/bench.log
for cfg in "${cfgs[@]}"; do
  if [ "$cfg" = "-" ]; then e=""; else e="$cfg"; fi
  echo "== $cfg" | tee -a $out/fixed_base.log
  for r in 1 2; do env $e python3 tools/fixed_base_timing.py 21 8 2>&1 | tail -1 | tee -a $out/fixed_base.log; done
done
