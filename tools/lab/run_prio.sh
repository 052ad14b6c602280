#!/bin/bash
# Round 6: wave priorities inside a proof.  The witness MSMs' accumulations run at priority 0, their short kernels at 3, the
# polynomial chain's kernels at K16_CHAIN_PRIO (3 until this experiment, 1 since); B2's (G2) chain ends ~0.6 ms after the
# polynomial chain and the H accumulation starts when it does.  Variants = alt builds of the library, made first with
#   cd keyless-zk-proofs_amd && for p in 0 2 3; do ./build_alt.sh chain$p "-DK16_CHAIN_PRIO=$p" ntt prover; done
#   ./build_alt.sh g2prio3 "-DK16_G2_ACC_PRIO=3" msm_g2;  ./build_alt.sh g2prio2 "-DK16_G2_ACC_PRIO=2" msm_g2
# (profiles/r06/ab_wave_priorities.log was taken when the default was still 3: its "default" row is today's chain3.)
out=${1:-gpurun_out/r6_prio}
mkdir -p "$out"
P=keyless-zk-proofs_amd/alt
variants=("" "chain3" "chain2" "chain0" "g2prio3" "g2prio2")
for r in 1 2 3 4 5; do
  for v in "${variants[@]}"; do
    if [ -z "$v" ]; then L=keyless-zk-proofs_amd/libk16.so; else L=$P/libk16_$v.so; fi
    K16_LIB_PATH=$L python3 tools/bench_proof.py --proofs 60 --no-stats --random-rs --concurrent 2 2>/dev/null | python3 -c "
import json,sys
l=[json.loads(x) for x in sys.stdin.read().splitlines() if x.startswith('{')]
print('%-18s p50 %.3f ms  p99 %.2f  %.1f proofs/s | two provers %.1f proofs/s' % ('${v:-default}', l[0]['p50_ms'], l[0]['p99_ms'], l[0]['value'], l[1]['value']))"
  done
done | tee "$out/ab_latency.log"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for v in "${variants[@]}"; do
  if [ -z "$v" ]; then L=keyless-zk-proofs_amd/libk16.so; else L=$P/libk16_$v.so; fi
  rm -rf /tmp/k16_tl; K16_LIB_PATH=$L rocprofv3 --kernel-trace --output-format csv -d /tmp/k16_tl -- python3 tools/bench_proof.py --proofs 6 --no-stats > /dev/null 2>&1
  { echo "variant: ${v:-default}"; python3 tools/proof_timeline.py /tmp/k16_tl 3; } > "$out/proof_timeline_${v:-default}.txt"
done
