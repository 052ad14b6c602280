#!/bin/bash
export K16_BENCH_NO_COLD=1 K16_BENCH_NO_HOST_LEG=1
A=$GRAFT_REPO_ROOT/keyless-zk-proofs_amd/alt
REPS=${REPS:-3} AB_ARGS="--steps 10 --warmup 3 --proofs 30 --no-cpu-baseline" tools/ab_bench.sh /tmp/ab "$@"
